/* TEST INFRASTRUCTURE ONLY — CPU restatement ("oracle") of the YAIK per-tile hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and
 * only as the checker / reported CPU baseline.  The product path (yaik_amd/, include/yaik_hip.h)
 * never links, imports or calls it.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit against the unmodified
 * reference sources compiled by oracle/Makefile (oracle/_ref/ref_driver) in
 * tests/test_oracle_vs_reference.py, and against the committed vectors in tests/golden/ that the
 * same binary produced (generator: tests/golden/make_golden.py).  The reference ships no tests or
 * fixtures of its own.  Since round 2 this includes yko_dec_mask (Decompress1BitTiled,
 * decoder/YAIK_Mipmap.cpp) and yko_image_builder (internal_imageBuilderFunc, decoder/YAIK_DefaultCallback.cpp):
 * both translation units are compiled into ref_driver (blobs dec_mask, dec_rgb_out, dec_rgba_out).
 */
#ifndef YAIK_ORACLE_H
#define YAIK_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct yko_enc yko_enc;   /* EncoderContext hot-path state (encoder/EncoderContext.h:273-323) */
typedef struct yko_dec yko_dec;   /* YAIK_Instance decode buffers (include/YAIK_private.h:26-54)      */

/* ---- encoder side ------------------------------------------------------------------------- */
/* planes: nPlanes (3 or 4) pointers to w*h int32, row-major (Plane, encoder/framework.h:74-127). Copied. */
yko_enc* yko_enc_create(int w, int h, int nPlanes, const int32_t* const* planes);
void     yko_enc_destroy(yko_enc* e);

/* EncoderContext::MipPrefilter (EncoderContext.cpp:1257-1427) incl. quadRecursion (:357-430).
 * Returns 1 if a 'MIPM' chunk would be written (bbox shrank), 0 if not, <0 on unsupported geometry
 * (the recursion is only well defined for w == h == 2^k). */
int yko_mip_prefilter(yko_enc* e);
/* out[0..3] = boundX0,boundY0,boundX1,boundY1 ; out[4] = mipMapTileSize ; out[5] = remainingPixels ;
 * out[6..9] = tile bbox x,y,w,h of the chunk (valid when a chunk is written) */
void yko_get_bounds(const yko_enc* e, int32_t out[10]);
const uint8_t* yko_mip_bitmap(const yko_enc* e, int* nBytes);

/* EncoderContext::FittingQuadSmooth (EncoderContext.cpp:3710-4363). planeBit: bit0/1/2 = srcA/B/C non-NULL.
 * Returns TileDone. Raw bitmap / corner stream of THIS call are readable until the next call. */
int yko_fitting_quad_smooth(yko_enc* e, int rejectFactor, int planeBit, int tileShiftX, int tileShiftY);
const uint8_t* yko_last_bitmap(const yko_enc* e, int* nBytes);
const uint8_t* yko_last_rgb_stream(const yko_enc* e, int* nBytes);   /* CompressF(Round6(corner),250) bytes */

/* state planes, all w*h bytes (0 / 255), except mappedRGB (w+1)*(h+1) */
const uint8_t* yko_smooth_map(const yko_enc* e);
const uint8_t* yko_mipmap_mask(const yko_enc* e);
const uint8_t* yko_map_smooth_tile(const yko_enc* e, int plane);
const int32_t* yko_preview(const yko_enc* e, int plane);            /* testOutput planes (:4096-4104) */

/* EncoderContext::DynamicTileEncode (EncoderContext.cpp:4365-4602) with GetTileEncode_Y (:1214),
 * Plane::GetMinMax_Y (Plane.cpp:489), GetTileDynamic_Y (:747), DynamicTile::buildTable (:625).
 * dst (w*h int32, caller-initialised) receives decoded values of valid pixels. Returns number of tile defs. */
int yko_dynamic_tile_encode(yko_enc* e, int mode3BitOnly, int plane, int32_t* dst);
const uint16_t* yko_last_tile_defs(const yko_enc* e, int* nDefs);
const uint8_t*  yko_last_nibbles(const yko_enc* e, int* nBytes, int* nNibbles);

/* EncoderContext::DynamicTileCompressor (EncoderContext.cpp:8398-8522): live 1-D range path for one plane,
 * map = mapSmoothTile[plane]. Appends to internal pixel/type streams (like the reference's globals).
 * debugOut (w*h int32, nullable) receives the per-pixel reconstruction the reference writes to `debug`. */
int yko_dynamic_tile_compressor(yko_enc* e, int plane, int32_t* debugOut);
const uint8_t* yko_1d_pix_stream(const yko_enc* e, int* nBytes);
const uint8_t* yko_1d_type_stream(const yko_enc* e, int* nBytes);

/* DynamicTile::buildTable for one (min,max): lut[6][16] in mode order Lin4,Exp4,Log4,Lin3,Exp3,Log3; returns base7Bit | distance6Bit<<8 */
int yko_build_table(int minV, int maxV, int32_t lut[96]);
/* the 2x(16+8) powf curve constants buildTable uses, order: exp4[16], log4[16], exp3[8], log3[8] */
void yko_curve_constants(float out[48]);

/* scalar helpers (EncoderContext.cpp:3183-3207) */
int yko_round6(int v);
int yko_round6p(int v);
int yko_compress_f(int v, int rate);
int yko_uncompress_f(int v, int rate);
/* PaletteFullRangeRemapping (decoder/YAIK_GenericFunctions.cpp:128-137), in place */
void yko_palette_remap(uint8_t* data, int n, int originalRange);

/* PaletteCompressor (EncoderContext.cpp:3259-3502) on a raw corner stream; keeps the reference's persistent
 * code table inside `e`, so call it once per pass in pass order like FittingQuadSmooth does (:4279). Returns bytes. */
int yko_palette_compress(yko_enc* e, const uint8_t* input, int size);
const uint8_t* yko_last_palette(const yko_enc* e, int* nBytes);
/* PaletteDecompressor (decoder/YAIK_GenericFunctions.cpp:139-241); input readable for inputSize+384 bytes. */
int yko_palette_decompress(const uint8_t* input, int inputSize, uint8_t* output, int outputSize, int colorCompression);

/* ---- decoder side ------------------------------------------------------------------------- */
yko_dec* yko_dec_create(int w, int h);
void     yko_dec_destroy(yko_dec* d);
/* DecompressGradient{16x16,16x8,8x16,8x8,8x4,4x8,4x4} (decoder/YAIK_Gradient.cpp:28-1418), RGB (planeBit 7).
 * rgb = de-quantised corner stream (after PaletteFullRangeRemapping). Returns bytes of rgb consumed. */
int yko_dec_gradient(yko_dec* d, int tileShiftX, int tileShiftY, const uint8_t* bitmap, int bitmapBytes, const uint8_t* rgb, int rgbBytes);
/* DecompressGradient4x4 with planeBit 1..6 (R, G, RG, B, RB, GB; decoder/YAIK_Gradient.cpp:1208-1226, :1420-2732); the masks must have
 * been split (yko_dec_split_masks) as YAIK_API.cpp:875-877 does for every 'GTIL' chunk whose plane field is not 7.
 * consistentMarks = 0 reproduces the reference's tile4x4Mask marking of these loops, defects included (no marks from R / G / B,
 * B marks of GB / RB at a wrong offset): that is what the pinned vectors hold; 1 marks each present plane's own mask, which is what
 * the ENCODER's per-plane coverage (and therefore Decompress1D behind it) assumes. */
int yko_dec_gradient_planes(yko_dec* d, int planeBit, int consistentMarks, const uint8_t* bitmap, int bitmapBytes, const uint8_t* rgb, int rgbBytes);
/* UpdateTileAndRGBMask (decoder/YAIK_API.cpp:530-544) */
void yko_dec_split_masks(yko_dec* d);
/* Decompress1D (decoder/YAIK_3DTile.cpp:24-240) for one plane; advances the two cursors. */
int yko_dec_1d(yko_dec* d, int plane, const uint8_t* type, int* typePos, const uint8_t* pix, int* pixPos, int compressionRange);
/* Decompress1BitTiled (decoder/YAIK_Mipmap.cpp:23-154), mipmapLevel 4 only. out = (bw*bh*256)/8 bytes swizzled mask. */
int yko_dec_mask(const uint8_t* bits, int tileBBoxW, int tileBBoxH, uint8_t* out);
/* internal_imageBuilderFunc (decoder/YAIK_DefaultCallback.cpp:24-191), w/h multiples of 8; alpha nullable (then 3 B/pixel).
 * With alpha the reference's RGBA branch is reproduced as it executes (RGB triples + one alpha byte per row, see the .c). */
int yko_image_builder(const uint8_t* planes, int planeSize, int w, int h, const uint8_t* alpha, int strideA, uint8_t* out, int stride);

const uint8_t* yko_dec_planes(const yko_dec* d, int* planeSize);     /* R|G|B, 8x8-tiled u8 */
const uint8_t* yko_dec_tile4x4(const yko_dec* d, int* sizePerPlane);
const uint8_t* yko_dec_map_rgb(const yko_dec* d, int* nBytes);
const uint8_t* yko_dec_map_rgb_mask(const yko_dec* d, int* sizePerPlane);

/* ---- (f)4 3-D LUT tile search (yaik_oracle_lut.c) ------------------------------------------------------------
 * Load3DPattern + Set3DPointCloud (EncoderContext.cpp:7851-7917, :4744-4814): count <= 64 points with 6-bit coordinates; returns the
 * pattern index.  factors: s16 [4 (6,5,4,3 bit)][3 (x,y,z)][64]; positions(step): u8 [64^3] nearest entry at 6 - step bits. */
int yko_lut_load(yko_enc* e, const uint8_t* r, const uint8_t* g, const uint8_t* b, int count);
int yko_lut_count(const yko_enc* e);
const int16_t* yko_lut_factors(const yko_enc* e, int pattern);
const int32_t* yko_lut_distance_field(const yko_enc* e, int pattern);
const uint8_t* yko_lut_positions(const yko_enc* e, int pattern, int step);
/* StartCorrelationSearch (:7316) / Correlation3DSearch (:6245, with computeValues3D :5807) for one tile shape; the streams accumulate over
 * the calls like the reference's corr3D_* buffers.  map: 0..5 = 16x8, 8x16, 8x8, 8x4, 4x8, 4x4.  indices(bits): raw entry numbers (the
 * chunk stores them x 3, :7526). */
void yko_lut_start(yko_enc* e);
int yko_lut_search(yko_enc* e, int tileShiftX, int tileShiftY);
const uint16_t* yko_lut_tile_types(const yko_enc* e, int* n);
const uint8_t* yko_lut_colors(const yko_enc* e, int* n);
const uint8_t* yko_lut_indices(const yko_enc* e, int bits, int* n);
const uint8_t* yko_lut_map(const yko_enc* e, int which, int* n);
const int32_t* yko_lut_preview(const yko_enc* e, int plane);
/* the decoder's 'LUL0' LUT file for the loaded bank (RegisterAndCreate3DLut :7820-7847 / BinarySave3D :5452); out == NULL: size */
int yko_lut_file(const yko_enc* e, uint8_t* out, int cap);
/* YAIK_AssignLUT's tables (decoder/YAIK_API.cpp:133-415) + Tile3D_16x8 .. 4x4 (decoder/YAIK_3DTile.cpp:244-2140) on the streams of a '3DTL'
 * chunk: maps in chunk order (16x8, 8x16, 8x8, 8x4, 4x8, 4x4), colours after PaletteFullRangeRemapping, indices as stored (x 3).
 * consumed: bytes of tile types, colours, 3/4/5/6-bit indices.  Runs before the masks are split (the chunk precedes '1DTL'). */
int yko_dec_lut3d(yko_dec* d, const uint8_t* lutFile, int lutBytes, const uint8_t* const maps[6], const uint16_t* tiles, int nTiles,
                  const uint8_t* colors, const uint8_t* const idx[4], int consumed[6]);

#ifdef __cplusplus
}
#endif
#endif
