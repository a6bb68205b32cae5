// Decoder drop-in for the tile hot path: the C API of the reference's include/YAIK.h (YAIK_Init .. YAIK_GetErrorCode,
// lines 52-143) with the chunk loops executed on an MI355X through include/yaik_hip.h.  Programs keep including the
// reference's own YAIK.h and link this library instead of the reference decoder: the declarations below repeat that header's
// types with identical layout, names and enumerator values so both sides agree on the ABI.
//
// On the path: file header, 'MIPM' (mask decode), 'GTIL' (gradient decode, all planes or a plane subset), '3DTL' (3-D LUT tiles, after
// YAIK_AssignLUT), '1DTL' (range decode), terminator, the default image builder (RGB888 / RGBA8888 rows at outputImageStride) and the
// custom-builder callback (8x8-tiled planes).
// Off the path (SURVEY §8 out of scope), reported through the sticky error code instead of decoded: 'ALPM' alpha value chunks
// (YAIK_ALPHA_UNSUPPORTED_YET); a '3DTL' chunk without an assigned LUT gives YAIK_INVALID_LUT.  Images whose sides are not multiples of 16 are refused
// (YAIK_INVALID_HEADER): the reference's own loops mis-stride there (decoder/YAIK_Gradient.cpp:15).
#pragma once
#include <stddef.h>
#include <stdint.h>

typedef void* YAIK_LIB;
typedef void* YAIK_INSTANCE;

typedef void* (*YAIK_allocFunc)(void* customContext, size_t size);
typedef void  (*YAIK_freeFunc)(void* customContext, void* address);     // must accept NULL
struct YAIK_SMemAlloc {
    YAIK_SMemAlloc() : customAlloc(0), customFree(0), customContext(0) {}
    YAIK_allocFunc customAlloc;
    YAIK_freeFunc  customFree;
    void*          customContext;
};

struct YAIK_SDecodedImage;
struct YAIK_SCustomDataSource {                // 8x8-tiled u8 planes R,G,B (+ linear alpha or NULL) and their strides
    uint8_t *planeR, *planeG, *planeB, *planeA;
    int32_t strideR, strideG, strideB;         // bytes to the next row of tiles
    int32_t strideA;                           // bytes to the next line
};
typedef void (*imageBuilderFunc)(struct YAIK_SDecodedImage* userInfo, struct YAIK_SCustomDataSource* sourceImageInternal);

struct YAIK_SDecodedImage {
    uint16_t         width, height;            // filled by YAIK_DecodeImagePre
    bool             hasAlpha;                 // filled by YAIK_DecodeImagePre
    imageBuilderFunc customImageOutput;        // Pre installs the default builder; the user may replace it before YAIK_DecodeImage
    void*            userContextCustomImage;
    YAIK_SMemAlloc   userMemoryAllocator;      // Pre installs malloc/free; the user may replace it.  HOST memory of a decode only: the planes,
                                                // corner lattice, masks and streams of this implementation live in HBM (hipMalloc, owned by the
                                                // decode slot's device handle) and never come from this allocator
    uint8_t*         outputImage;              // user buffer, required by YAIK_DecodeImage
    int32_t          outputImageStride;
    bool             hasAlpha1Bit;
    YAIK_INSTANCE    internalTag;              // owned by the library between Pre and Decode
};

enum YAIK_ERROR_CODE {
    YAIK_NO_ERROR = 0, YAIK_INVALID_LIBRARYCTX, YAIK_MALLOC_FAIL, YAIK_INVALID_CONTEXT_COUNT, YAIK_INIT_FAIL,
    YAIK_RELEASE_EMPTY_LIBRARY, YAIK_INVALID_STREAM, YAIK_INVALID_HEADER, YAIK_NO_EMPTYDECODE_SLOT, YAIK_DECIMG_INVALIDCTX,
    YAIK_DECIMG_DIFFSTREAM, YAIK_DECIMG_BUFFERNOTSET, YAIK_INVALID_CONTEXT_MEMALLOCATOR, YAIK_INVALID_DECOMPRESSION, YAIK_INVALID_LUT,
    YAIK_DECOMPRESSION_CREATE_FAIL, YAIK_INVALID_MIPMAP_LEVEL, YAIK_ALPHA_FORMAT_IMPOSSIBLE, YAIK_INVALID_ALPHA_FORMAT,
    YAIK_ALPHA_UNSUPPORTED_YET, YAIK_INVALID_TAG_ID, YAIK_INVALID_PLANE_ID,
};

YAIK_LIB        YAIK_Init(uint8_t maxDecodeThreadContext, YAIK_SMemAlloc* libraryMemAllocator);
void            YAIK_AssignLUT(YAIK_LIB lib, uint8_t* lutData, uint32_t lutDataLength);     // the 3-D LUT file ('LUL0'); needed before a stream with a '3DTL' chunk
void            YAIK_Release(YAIK_LIB lib);
bool            YAIK_DecodeImagePre(YAIK_LIB lib, void* sourceStreamAligned, uint32_t streamLength, YAIK_SDecodedImage* getUserInfo);
bool            YAIK_DecodeImage(void* sourceStreamAligned, uint32_t streamLength, YAIK_SDecodedImage* context);
YAIK_ERROR_CODE YAIK_GetErrorCode();

// extension (not in the reference): HIP device used by the decode slots created by the next YAIK_Init (default 0)
void            YAIK_SetDevice(int device);
// extension: how 'GTIL' chunks for one or two planes (HeaderGradientTile::plane 1..6) mark tile4x4Mask.  0 (default) = exactly what the
// reference's DecompressGradient4x4R/G/B/RG/GB/RB loops do (R/G/B leave the mask alone, GB/RB put the B marks at tile4x4Mask +
// tile4x4MaskSize/2; decoder/YAIK_Gradient.cpp:1420-2732), after which the reference's own Decompress1D reads a different number of
// tiles than the encoder wrote; 1 = every pass marks the planes it filled, which decodes such streams correctly.
void            YAIK_SetPartialPlaneMarks(int consistent);
