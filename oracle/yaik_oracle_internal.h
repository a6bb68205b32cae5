/* TEST INFRASTRUCTURE ONLY.  The encoder state shared by the translation units of liboracle.so. */
#pragma once
#include "yaik_oracle.h"
#include <stddef.h>

struct yko_lut_state;
void yko_lut_free(struct yko_lut_state* s);

struct yko_enc {
    int w, h, nPlanes;
    int32_t* plane[4];
    /* EncoderContext state (EncoderContext.h:273-323), kept as bytes 0/255 */
    uint8_t* mipmapMask;            /* NULL until CheckMipmapMask / MipPrefilter */
    uint8_t* smoothMap;             /* NULL until first FittingQuadSmooth (:3739) */
    uint8_t* mapSmoothTile[3];
    uint8_t* mappedRGB[3];          /* (w+1)*(h+1) */
    int32_t* preview[3];
    int boundX0, boundY0, boundX1, boundY1, mipMapTileSize, remainingPixels;
    int tileBBox[4];
    uint8_t* mipBitmap; int mipBitmapBytes;
    uint8_t* lastBitmap; int lastBitmapBytes;
    uint8_t* lastRgb; int lastRgbBytes;
    uint16_t* tileDefs; int nDefs;
    uint8_t* nibbles; int nNibbles;
    uint8_t* pix1d; int nPix1d; size_t capPix1d;
    uint8_t* type1d; int nType1d; size_t capType1d;
    /* PaletteCompressor's process-global code table CodeRGB/CodeCount (EncoderContext.cpp:3216-3217): only the
     * count is reset per call, stale rows stay and are still matched by FindCodeBook (:3248-3255). */
    int (*codeRGB)[4]; int codeCount, codeCap;   /* ref, dr, dg, db */
    uint8_t* lastPalette; int lastPaletteBytes;
    struct yko_lut_state* lut;      /* (f)4 3-D LUT search state (yaik_oracle_lut.c), NULL until the first pattern is loaded */
};


struct yko_dec {
    int w, h, tileW, tileH, planeSize;
    uint8_t* planes;        /* R|G|B, 8x8-tiled u8 (include/YAIK.h:205-224) */
    int strideRGBMap, lattice;
    uint8_t* mapRGB;        /* lattice*3 */
    uint8_t* mapRGBMask; int sizeMapMask;
    uint8_t* tile4x4Mask; int tile4x4MaskSize, stride4;
    int singleRGB;          /* masks still in single-plane form (YAIK_Instance::singleRGB) */
};

