// Chunk framing of the tile hot path: what the reference's passes fwrite to `outFile` after handing their raw streams to
// PaletteCompressor / ZStd.  One function per chunk, each citing the reference writer it mirrors.  Payload bytes produced by
// ZStd differ between library versions; headers, stream order, padding and the decompressed payloads are the contract.
#pragma once
#include <cstdio>
#include <string>
#include <vector>
#include "framework.h"

namespace yaikchunk {
bool writeFileHeader(FILE* f, int width, int height, bool hasAlpha);                    // EncoderContext.cpp:9007-9016
bool writeEndOfFile(FILE* f);                                                           // :9779-9781
// 'MIPM' (:1367-1396): bbox in 16x16 tiles, 1 bit per tile, not compressed
bool writeMipmap(FILE* f, const int tileBBox[4], int mipmapLevel, const u8* bits, size_t nBytes);
// 'GTIL' (:4239-4347).  Returns 1 = chunk written, 0 = nothing to write (no accepted tile / empty colour stream), -1 = error
int  writeGradientTile(FILE* f, int imgW, int imgH, int tileShiftX, int tileShiftY, const u8* bitmap, size_t bitmapBytes,
                       u8* rgbStream, size_t rgbBytes, int colorCompression, int planeBit, std::string& err);
// 'PLNT' (:4516-4589)
bool writePlaneTile(FILE* f, const BoundingBox& constraint, const u16* defs, size_t nDefs, const u8* idx, size_t idxBytes,
                    int planeType, bool halfX, bool halfY, std::string& err);
// '3DTL' (EndCorrelationSearch, :7366-7678): the six tile maps (16x8, 8x16, 8x8, 8x4, 4x8, 4x4), tile types, box colours (CompressF'd here
// with colorCompression), the 3/4/5/6-bit entry numbers (stored x 3).  The header keeps map sizes in 16 bits: larger maps are refused.
struct Tile3DStreams { const u8* map[6]; size_t mapBytes[6]; const u16* tileType; size_t nTiles; const u8* color; const u8* idx[4]; size_t nIdx[4]; };
bool writeTile3D(FILE* f, const Tile3DStreams& s, int colorCompression, int component, std::string& err);
// '1DTL' (GenerateDynamicTileChunk, :8524-8576): type stream first, then the pixel stream
bool writeTile1D(FILE* f, const u8* pix, size_t pixBytes, const u8* type, size_t typeBytes, int compressionColor,
                 int compressionRange, std::string& err);
// The two compressed chunk kinds in split form, for a threaded entropy stage (EncoderContext::ConvertHotPathBegin): the streams are
// compressed wherever (compressStream = CompressStream, :3692-3708: plain ZSTD_compress at the given level), the chunks are emitted
// in file order.  writeGradientTile / writeTile1D are these two steps back to back.
bool compressStream(const void* src, size_t n, int level, std::vector<u8>& out, std::string& err);
bool gradientTileHasChunk(int imgW, int imgH, int tileShiftX, int tileShiftY, const u8* bitmap, size_t rgbBytes);
bool emitGradientTile(FILE* f, int imgW, int imgH, int tileShiftX, int tileShiftY, const u8* bitmap, size_t rgbBytes, u32 paletteBytes,
                      const std::vector<u8>& zBitmap, const std::vector<u8>& zRgb, int colorCompression, int planeBit, std::string& err);
bool emitTile1D(FILE* f, size_t pixBytes, size_t typeBytes, const std::vector<u8>& zPix, const std::vector<u8>& zType,
                int compressionColor, int compressionRange, std::string& err);
// extent of the set bits of a swizzled gradient bitmap in pixels: {minX, minY, maxX, maxY} (:3798-3799, :4039-4042)
void gradientExtent(int imgW, int imgH, int tileShiftX, int tileShiftY, const u8* bitmap, int out[4]);
}
