// ZStd entropy stage on host cores.  The reference vendors zstd 1.3.4 (external/zstd) and calls ZSTD_compress at levels 18/21
// (EncoderContext.cpp:3692-3708, 4519, 4533, 8539, 8548) and ZSTD_decompressDCtx (decoder/YAIK_API.cpp:503-519).  Here the
// system's libzstd.so.1 is loaded at run time (the image ships the library but not its header); any zstd >= 1.0 produces
// frames the reference's decoder reads, and the compressed bytes are not part of the parity contract (lossless stage).
#pragma once
#include <cstddef>

namespace yaikzstd {
bool   available();                                   // false: libzstd.so.1 could not be loaded (message via lastError())
const char* lastError();
size_t compressBound(size_t srcSize);
// returns the compressed size, or 0 on error
size_t compress(void* dst, size_t dstCapacity, const void* src, size_t srcSize, int level);
// returns true when exactly expectedSize bytes came out
bool   decompress(void* dst, size_t expectedSize, const void* src, size_t srcSize);
// expands a frame of unknown size into a buffer that is large enough; returns false on error or overflow
bool   decompressAny(void* dst, size_t dstCapacity, const void* src, size_t srcSize, size_t* outSize);
}
