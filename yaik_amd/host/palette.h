// Delta / code-book codec of the gradient corner-colour stream: the entropy-stage partner of ZStd, run on host cores.
//   PaletteCompressor    encoder/EncoderContext.cpp:3259-3502 (registerCodeBook :3231, FindCodeBook :3248, compCode :3222)
//   PaletteDecompressor  decoder/YAIK_GenericFunctions.cpp:139-241 (+ PaletteFullRangeRemapping :128-137)
// Same signatures and return conventions as the reference.  The code table is process-global like the reference's
// CodeRGB/CodeCount (:3216-3217): FindCodeBook scans rows 0..63 whether or not the current call filled them, so rows left
// over from an earlier call take part in the match — kept, because the emitted bytes depend on it.  (Global per thread: see palette.cpp.)
#pragma once
#include "framework.h"

bool PaletteCompressor(u8* input, int size, u8* output, u32* maxSizeInOut);
bool PaletteDecompressor(u8* input, int inputSize, int inputBufferSize, u8* output, int outputSize, u8 colorCompression);
void PaletteFullRangeRemapping(u8* data, int size, u8 originalRange);
void PaletteResetCodeBook();                            // test hook: forget the rows of earlier calls (a fresh process)
