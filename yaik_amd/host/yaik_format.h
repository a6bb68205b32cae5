// On-disk layout of a .yaik stream, as the reference writes and reads it (include/YAIK_private.h:85-352 in KLab/YAIK).
// Only the chunks of the tile hot path are described: 'MIPM', 'GTIL', 'PLNT', '1DTL', '3DTL'.  All structs are plain little-endian C
// layouts with the natural padding the reference's compilers give them; the static_asserts pin the sizes the reference's
// decoder steps over with `&pHeader[1]`.  Padding bytes and `HeaderGradientTile::version` are never initialised by the
// reference (they carry stack garbage there); this writer stores zeros.
#pragma once
#include <cstdint>
#include "framework.h"

namespace yaikfmt {

static const u32 TAG_FILE     = 0x4b494159u;   // 'Y','A','I','K'  (encoder/EncoderContext.cpp:9010-9013)
static const u32 TAG_MIPMAP   = 0x4d50494du;   // 'M','I','P','M'  (decoder/YAIK_API.cpp:560)
static const u32 TAG_GRADTILE = 0x4c495447u;   // 'G','T','I','L'  (:562)
static const u32 TAG_TILE1D   = 0x4c544431u;   // '1','D','T','L'  (:564)
static const u32 TAG_TILE3D   = 0x4c544433u;   // '3','D','T','L'  (EndCorrelationSearch, EncoderContext.cpp:7589-7593; reader decoder/YAIK_API.cpp:999)
static const u32 TAG_PLANE    = 0x544e4c50u;   // 'P','L','N','T'  (written by DynamicTileEncode, EncoderContext.cpp:4541-4545; no reader)
static const u32 TAG_END      = 0xDEADBEEFu;   // terminator (EncoderContext.cpp:9779-9781)

struct FileHeader {                             // YAIK_private.h:96-105
    u32 tag;
    u16 version, width, height, infoMask;       // infoMask bit 0: alpha channel present
};
struct HeaderBase { u32 tag, length; };         // :107-110; length = payload bytes after this header, rounded up to 4

struct MipmapHeader {                           // :112-118
    BoundingBox bbox;                           // in 16x16 tiles
    u32 streamSize;                             // not written by the reference
    u8 version, mipmapLevel;
};
struct HeaderGradientTile {                     // :172-211
    BoundingBox bbox;
    u32 streamBitmapSize, streamRGBSizeZStd, streamRGBSizeCustomCompressor, streamRGBSizeUncompressed;
    u8 colorCompression, version, format, plane;
};
struct PlaneTile {                              // :288-299
    BoundingBox bbox;
    u32 streamSizeTileMap, streamSizeTileStream, expectedSizeTileStream;
    u8 version, format;
};
struct Header1D {                               // :341-350
    u32 streamPixelBit, streamPixelUncmp, streamTypeCnt, streamTypeUncmp;
    u8 compressionColor, compressionRange, version;
};

struct HeaderTile3D {                           // :302-332
    u32 streamColorCnt, streamTypeCnt, stream3BitCnt, stream4BitCnt, stream5BitCnt, stream6BitCnt;
    u32 comprTypeSize, comprColorSize, compr3BitSize, compr4BitSize, compr5BitSize, compr6BitSize;
    u16 sizeT16_8Map, sizeT8_16Map, sizeT8_8Map, sizeT4_8Map, sizeT8_4Map, sizeT4_4Map;
    u16 sizeT16_8MapCmp, sizeT8_16MapCmp, sizeT8_8MapCmp, sizeT4_8MapCmp, sizeT8_4MapCmp, sizeT4_4MapCmp;
    u8 component, compressionRateColor;
};
struct LUTHeader { u8 lutH[4]; u8 version, entryCount; u8 padding_extension[2]; };      // :75-80, the decoder's LUT file
static_assert(sizeof(HeaderTile3D) == 76 && sizeof(LUTHeader) == 8, "3-D LUT chunk / file headers");
static_assert(sizeof(FileHeader) == 12 && sizeof(HeaderBase) == 8, "file framing");
static_assert(sizeof(MipmapHeader) == 16 && sizeof(HeaderGradientTile) == 28, "chunk headers");
static_assert(sizeof(PlaneTile) == 24 && sizeof(Header1D) == 20, "chunk headers");

// HeaderGradientTile::getSwizzleSize (YAIK_private.h:212-276): swizzle block and tiles per block for a tile shape
inline bool swizzleSize(int shiftX, int shiftY, u32& bigX, u32& bigY, u32& bitCount) {
    bigX = bigY = bitCount = 0;
    if ((shiftX == 4 || shiftX == 3) && (shiftY == 4 || shiftY == 3)) { bigX = 64; bigY = 64; }
    else if (shiftX == 3 && shiftY == 2) { bigX = 64; bigY = 32; }
    else if (shiftX == 2 && shiftY == 3) { bigX = 32; bigY = 64; }
    else if (shiftX == 2 && shiftY == 2) { bigX = 32; bigY = 32; }
    else return false;
    bitCount = (bigX >> shiftX) * (bigY >> shiftY);
    return true;
}

}  // namespace yaikfmt
