/* TEST INFRASTRUCTURE ONLY — see yaik_oracle.h.  Plain-C CPU restatement of the YAIK hot path.
 * Written from the behaviour of the reference (file:line cited per function); no reference code is
 * copied.  Deliberately simple and scalar: it is the checker, not the product.
 * Build: gcc -O2 -std=c11 -ffp-contract=off (float expression order matters in the range quantiser).
 */
#include "yaik_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------- */
#include "yaik_oracle_internal.h"

yko_enc* yko_enc_create(int w, int h, int nPlanes, const int32_t* const* planes) {
    if (w <= 0 || h <= 0 || nPlanes < 3 || nPlanes > 4) return NULL;
    yko_enc* e = (yko_enc*)calloc(1, sizeof *e);
    e->w = w; e->h = h; e->nPlanes = nPlanes;
    for (int p = 0; p < nPlanes; p++) {
        e->plane[p] = (int32_t*)malloc(sizeof(int32_t) * (size_t)w * h);
        memcpy(e->plane[p], planes[p], sizeof(int32_t) * (size_t)w * h);
    }
    return e;
}

void yko_enc_destroy(yko_enc* e) {
    if (!e) return;
    for (int p = 0; p < 4; p++) free(e->plane[p]);
    free(e->mipmapMask); free(e->smoothMap);
    for (int p = 0; p < 3; p++) { free(e->mapSmoothTile[p]); free(e->mappedRGB[p]); free(e->preview[p]); }
    free(e->mipBitmap); free(e->lastBitmap); free(e->lastRgb); free(e->tileDefs); free(e->nibbles);
    free(e->pix1d); free(e->type1d); free(e->codeRGB); free(e->lastPalette);
    yko_lut_free(e->lut);
    free(e);
}

/* Plane::GetPixelValue clamp-to-edge read (encoder/framework.h:116-121) */
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int px(const int32_t* p, int w, int h, int x, int y) {
    return p[clampi(x, 0, w - 1) + clampi(y, 0, h - 1) * w];
}

/* EncoderContext::CheckMipmapMask (EncoderContext.cpp:2784-2794) */
static void check_mipmap_mask(yko_enc* e) {
    if (!e->mipmapMask) {
        e->mipmapMask = (uint8_t*)malloc((size_t)e->w * e->h);
        memset(e->mipmapMask, 255, (size_t)e->w * e->h);
        e->boundX0 = 0; e->boundY0 = 0; e->boundX1 = e->w; e->boundY1 = e->h;
    }
}

/* ------------------------------------------------------------------------------------------- */
/* a9: alpha tile-reject.  quadRecursion (EncoderContext.cpp:357-430), maxMipLevel = 3 (:1275). */
typedef struct { yko_enc* e; int L, T, R, B; } quad_ctx;

static int quad_rec(quad_ctx* q, int depth, int x, int y) {
    yko_enc* e = q->e;
    depth--;
    int sq = 1 << depth, sqChild = sq >> 1;
    if (depth > 0) {
        int u = x + sqChild, v = y + sqChild;
        int allowDown = v < e->w;                 /* the reference tests the y child against the WIDTH (:373) */
        int a = quad_rec(q, depth, x, y), b, c, d;
        b = allowDown ? quad_rec(q, depth, x, v) : 1;
        if (u < e->w) {
            c = quad_rec(q, depth, u, y);
            d = allowDown ? quad_rec(q, depth, u, v) : 1;
        } else { c = 1; d = 1; }
        int res = a && b && c && d;
        if (res) {
            if (depth > 3) {                      /* node side >= 16: clear the mask (:396-414) */
                int sw = (x + sq > e->w) ? e->w - x : sq;
                int sh = (y + sq > e->h) ? e->h - y : sq;
                for (int yy = y; yy < y + sh; yy++) memset(e->mipmapMask + (size_t)yy * e->w + x, 0, (size_t)(sw > 0 ? sw : 0));
            }
        } else if (depth == 4) {                  /* kept 16x16 node grows the bounding box (:416-422) */
            if (q->L > x) q->L = x;
            if (q->T > y) q->T = y;
            if (q->R < x + sq) q->R = x + sq;
            if (q->B < y + sq) q->B = y + sq;
        }
        return res;
    }
    return e->plane[3][x + y * e->w] == 0;
}

int yko_mip_prefilter(yko_enc* e) {
    int w = e->w, h = e->h;
    free(e->mipmapMask);
    e->mipmapMask = (uint8_t*)malloc((size_t)w * h);
    memset(e->mipmapMask, 255, (size_t)w * h);
    free(e->mipBitmap); e->mipBitmap = NULL; e->mipBitmapBytes = 0;
    if (e->nPlanes != 4) {                         /* :1419-1426 */
        e->boundX0 = 0; e->boundY0 = 0; e->boundX1 = w; e->boundY1 = h;
        e->remainingPixels = w * h;
        return 0;
    }
    if (w != h || (w & (w - 1)) || w < 16) return -1;   /* recursion reads out of bounds otherwise */
    int maxSize = w > h ? w : h, lvl = 0;
    for (unsigned v = (unsigned)maxSize; v >>= 1;) lvl++;  /* log2ui (:43-50) */
    quad_ctx q = { e, 9999999, 9999999, -1, -1 };
    quad_rec(&q, lvl + 1, 0, 0);
    e->boundX0 = q.L; e->boundX1 = q.R; e->boundY0 = q.T; e->boundY1 = q.B;
    e->mipMapTileSize = 16;
    if (e->boundX0 != 0 || e->boundY0 != 0 || e->boundX1 != w || e->boundY1 != h) {
        int bx0 = e->boundX0 >> 4, bx1 = e->boundX1 >> 4, by0 = e->boundY0 >> 4, by1 = e->boundY1 >> 4;
        int tw = bx1 - bx0, th = by1 - by0;
        int sizeByte = ((tw * th) + 7) / 8;
        if (sizeByte < 0) sizeByte = 0;
        e->mipBitmap = (uint8_t*)calloc((size_t)sizeByte + 1, 1);
        e->mipBitmapBytes = sizeByte;
        e->remainingPixels = 0;
        int bitPos = 0;
        for (int y = 0; y < th; y++) for (int x = 0; x < tw; x++) {       /* :1317-1327 */
            if (e->mipmapMask[(size_t)((x + bx0) * 16) + (size_t)((y + by0) * 16) * w]) {
                e->mipBitmap[bitPos >> 3] |= (uint8_t)(1 << (bitPos & 7));
                e->remainingPixels += 256;
            }
            bitPos++;
        }
        e->tileBBox[0] = bx0; e->tileBBox[1] = by0; e->tileBBox[2] = tw; e->tileBBox[3] = th;
        return 1;
    }
    memset(e->mipmapMask, 255, (size_t)w * h);                             /* :1401 */
    e->remainingPixels = e->boundX1 * e->boundY1;
    return 0;
}

void yko_get_bounds(const yko_enc* e, int32_t out[10]) {
    out[0] = e->boundX0; out[1] = e->boundY0; out[2] = e->boundX1; out[3] = e->boundY1;
    out[4] = e->mipMapTileSize; out[5] = e->remainingPixels;
    for (int i = 0; i < 4; i++) out[6 + i] = e->tileBBox[i];
}
const uint8_t* yko_mip_bitmap(const yko_enc* e, int* n) { *n = e->mipBitmapBytes; return e->mipBitmap; }

/* ------------------------------------------------------------------------------------------- */
/* scalar helpers (EncoderContext.cpp:3183-3207) */
int yko_round6(int v) { int r = v >> 2; return (r << 2) | (r >> 4); }
int yko_round6p(int v) { v++; if (v > 255) v = 255; int r = v >> 2; return (r << 2) | (r >> 4); }
int yko_compress_f(int v, int rate) { return (v * rate + 127) / 255; }
int yko_uncompress_f(int v, int rate) { int inv = rate ? (255 << 16) / rate : (255 << 16); return (v * inv) >> 16; }

void yko_palette_remap(uint8_t* data, int n, int originalRange) {
    int inv = originalRange ? ((255 << 16) / originalRange) : (255 << 16);
    for (int i = 0; i < n; i++) data[i] = (uint8_t)((data[i] * inv) >> 16);
}

/* HeaderGradientTile::getSwizzleSize (include/YAIK_private.h:212-276) */
static void swizzle_size(int sx, int sy, int* bigX, int* bigY, int* bitCount) {
    int X = 0, Y = 0;
    if (sx == 4 && (sy == 4 || sy == 3)) { X = 64; Y = 64; }
    else if (sx == 3 && (sy == 4 || sy == 3)) { X = 64; Y = 64; }
    else if (sx == 3 && sy == 2) { X = 64; Y = 32; }
    else if (sx == 2 && sy == 3) { X = 32; Y = 64; }
    else if (sx == 2 && sy == 2) { X = 32; Y = 32; }
    *bigX = X; *bigY = Y; *bitCount = (X >> sx) * (Y >> sy);
}

/* ------------------------------------------------------------------------------------------- */
/* a6: FittingQuadSmooth (EncoderContext.cpp:3710-4363) */
int yko_fitting_quad_smooth(yko_enc* e, int rejectFactor, int planeBit, int sx, int sy) {
    check_mipmap_mask(e);
    const int w = e->w, h = e->h;
    const int TX = 1 << sx, TY = 1 << sy;
    const int32_t* src[3];
    for (int n = 0; n < 3; n++) src[n] = (planeBit & (1 << n)) ? e->plane[n] : NULL;

    if (!e->smoothMap) {                                               /* :3739-3749 */
        e->smoothMap = (uint8_t*)calloc((size_t)w * h, 1);
        for (int n = 0; n < 3; n++) {
            e->mapSmoothTile[n] = (uint8_t*)calloc((size_t)w * h, 1);
            e->mappedRGB[n] = (uint8_t*)calloc((size_t)(w + 1) * (h + 1), 1);
            e->preview[n] = (int32_t*)calloc((size_t)w * h, sizeof(int32_t));
        }
    }
    int bigX, bigY, bitCount;
    swizzle_size(sx, sy, &bigX, &bigY, &bitCount);
    if (!bigX) return -1;
    int xBB = (w + bigX - 1) / bigX, yBB = (h + bigY - 1) / bigY;
    int sizeBitmap = (xBB * yBB * bitCount) >> 3;
    free(e->lastBitmap);
    e->lastBitmap = (uint8_t*)calloc((size_t)sizeBitmap + 1, 1);
    e->lastBitmapBytes = sizeBitmap;
    int streamW = (w / TX) + 1, streamH = (h / TY) + 1;
    free(e->lastRgb);
    e->lastRgb = (uint8_t*)malloc((size_t)streamW * streamH * 12 + 16);
    uint8_t* wr = e->lastRgb;
    int tileDone = 0;
    int* preT[3];
    for (int n = 0; n < 3; n++) preT[n] = (int*)malloc(sizeof(int) * 256);

    const int stepY = bigX / TX;
    for (int by = 0; by < yBB; by++) for (int bx = 0; bx < xBB; bx++) {          /* :3810-3812 */
        int posBlock = (by * xBB + bx) * bitCount;
        int ty = 0;
        for (int y = by * bigY; y < by * bigY + bigY; y += TY, ty++) {               /* :3816 */
            if (y >= h || (y + TY) > h) break;
            int pos = posBlock + ty * stepY;
            for (int x = bx * bigX; x < bx * bigX + bigX; x += TX, pos++) {          /* :3823 */
                if (x >= w || (x + TX) > w) break;
                int c[4][3], c6[4][3], cp[4][3];                                     /* TL,TR,BL,BR */
                for (int n = 0; n < 3; n++) {
                    if (src[n]) {
                        c[0][n] = px(src[n], w, h, x, y);       c[1][n] = px(src[n], w, h, x + TX, y);
                        c[2][n] = px(src[n], w, h, x, y + TY);  c[3][n] = px(src[n], w, h, x + TX, y + TY);
                    } else { c[0][n] = c[1][n] = c[2][n] = c[3][n] = 0; }
                    for (int k = 0; k < 4; k++) { c6[k][n] = yko_round6(c[k][n]); cp[k][n] = yko_round6p(c[k][n]); }
                }
                int allow = 1;                                                        /* :3871-3875 top-left pixel only */
                for (int n = 0; n < 3; n++) if (src[n] && e->mapSmoothTile[n][x + (size_t)y * w] != 0) allow = 0;
                if (!allow) continue;

                int rej[6] = { 0, 0, 0, 0, 0, 0 };   /* blendC, blendC6, blendCO, blendC6O, blendC6OExp, blendC6Exp */
                const int rounding = (1 << 19) - 1;
                for (int dy = 0; dy < TY; dy++) {
                    int tF = 1024 - dy * (1024 / TY), bF = 1024 - tF;                 /* weight4/8/16 tables :3735-3737 */
                    for (int dx = 0; dx < TX; dx++) {
                        int lF = 1024 - dx * (1024 / TX), rF = 1024 - lF;
                        for (int n = 0; n < 3; n++) {
                            int cur = src[n] ? src[n][(x + dx) + (size_t)(y + dy) * w] : 0;
                            int T  = c [0][n] * lF + c [1][n] * rF, B  = c [2][n] * lF + c [3][n] * rF;
                            int T6 = c6[0][n] * lF + c6[1][n] * rF, B6 = c6[2][n] * lF + c6[3][n] * rF;
                            int TP = cp[0][n] * lF + cp[1][n] * rF, BP = cp[2][n] * lF + cp[3][n] * rF;
                            int S = T * tF + B * bF, S6 = T6 * tF + B6 * bF, SP = TP * tF + BP * bF;
                            int blendC   = (S  + rounding) / (1024 * 1024), blendCO   = S  / (1024 * 1024);
                            int blendC6  = (S6 + rounding) / (1024 * 1024), blendC6O  = S6 / (1024 * 1024);
                            int blendC6E = (SP + rounding) / (1024 * 1024), blendC6OE = SP / (1024 * 1024);
                            preT[n][dx + dy * TX] = blendC6E;
                            if (abs(cur - blendC)    > rejectFactor) rej[0] = 1;
                            if (abs(cur - blendC6)   > rejectFactor) rej[1] = 1;
                            if (abs(cur - blendCO)   > rejectFactor) rej[2] = 1;
                            if (abs(cur - blendC6O)  > rejectFactor) rej[3] = 1;
                            if (abs(cur - blendC6OE) > rejectFactor) rej[4] = 1;
                            if (abs(cur - blendC6E)  > rejectFactor) rej[5] = 1;
                        }
                    }
                }
                if (rej[0] && rej[1] && rej[2] && rej[3] && rej[4] && rej[5]) continue;   /* :3998 */

                /* accept: corner-seen lattice (:4001-4021) */
                int enc[4][3];
                const int cxs[4] = { x, x + TX, x, x + TX }, cys[4] = { y, y, y + TY, y + TY };
                for (int n = 0; n < 3; n++) for (int k = 0; k < 4; k++) {
                    enc[k][n] = 1;
                    if (src[n]) {
                        uint8_t* m = &e->mappedRGB[n][cxs[k] + (size_t)cys[k] * (w + 1)];
                        enc[k][n] = *m; if (!*m) *m = 255;
                    }
                }
                e->lastBitmap[pos >> 3] |= (uint8_t)(1 << (pos & 7));                  /* :4026 */
                for (int dy = 0; dy < TY; dy++) for (int dx = 0; dx < TX; dx++) {      /* :4029-4037 */
                    size_t i = (x + dx) + (size_t)(y + dy) * w;
                    e->smoothMap[i] = 255; e->mipmapMask[i] = 0;
                    for (int n = 0; n < 3; n++) if (src[n]) { e->mapSmoothTile[n][i] = 255; e->preview[n][i] = preT[n][dx + dy * TX]; }
                }
                tileDone++;
                for (int k = 0; k < 4; k++) for (int n = 0; n < 3; n++)                 /* :4113-4132 */
                    if (src[n] && !enc[k][n]) *wr++ = (uint8_t)yko_compress_f(c6[k][n], 250);
            }
        }
    }
    for (int n = 0; n < 3; n++) free(preT[n]);
    e->lastRgbBytes = (int)(wr - e->lastRgb);
    return tileDone;
}

const uint8_t* yko_last_bitmap(const yko_enc* e, int* n) { *n = e->lastBitmapBytes; return e->lastBitmap; }
const uint8_t* yko_last_rgb_stream(const yko_enc* e, int* n) { *n = e->lastRgbBytes; return e->lastRgb; }
const uint8_t* yko_smooth_map(const yko_enc* e) { return e->smoothMap; }
const uint8_t* yko_mipmap_mask(const yko_enc* e) { return e->mipmapMask; }
const uint8_t* yko_map_smooth_tile(const yko_enc* e, int p) { return e->mapSmoothTile[p]; }
const int32_t* yko_preview(const yko_enc* e, int p) { return e->preview[p]; }

/* ------------------------------------------------------------------------------------------- */
/* a13: DynamicTile::buildTable (EncoderContext.cpp:625-699) with MinRange / DiffRange helpers (:587-623) */
typedef struct { int base7Bit, distance6Bit; int lut[6][16]; } dyn_table;

static void build_table(int min_, int max_, dyn_table* t) {
    if (min_ > 224) min_ = 224;
    if (max_ > 255) max_ = 255;
    int diff = max_ - min_;
    if (diff < 16) diff = 16;
    int r8 = min_ > 224 ? 224 : min_;
    int base = (r8 * 63 + 112) / 224;                    /* MinRangeEncode */
    int BN = (base * 224) / 63;                          /* MinRangeDecode */
    int d8 = diff < 32 ? 32 : diff;                      /* DiffRangeEncode */
    int scale = (255 - 32) - BN;
    int dist = ((d8 - 32) * 127 + (scale - 1)) / scale;
    int rangeDecode = (dist * scale) / 127 + 32;         /* DiffRangeDecode */
    t->base7Bit = base; t->distance6Bit = dist;
    float DistNormF = (float)rangeDecode;
    for (int i = 0; i < 16; i++) {
        float pos = i / 15.0f;
        float e4 = powf(pos, 1.4f), l4 = 1.0f - powf(1.0f - pos, 1.4f);
        float oL = pos * DistNormF, oE = e4 * DistNormF, oG = l4 * DistNormF;
        t->lut[0][i] = (int)(BN + oL); t->lut[1][i] = (int)(BN + oE); t->lut[2][i] = (int)(BN + oG);
    }
    for (int i = 0; i < 8; i++) {
        float pos = i / 7.0f;
        float e3 = powf(pos, 1.4f), l3 = 1.0f - powf(1.0f - pos, 1.4f);
        float oL = pos * DistNormF, oE = e3 * DistNormF, oG = l3 * DistNormF;
        t->lut[3][i] = (int)(BN + oL); t->lut[4][i] = (int)(BN + oE); t->lut[5][i] = (int)(BN + oG);
        t->lut[3][i + 8] = t->lut[4][i + 8] = t->lut[5][i + 8] = 0;
    }
}

int yko_build_table(int minV, int maxV, int32_t lut[96]) {
    dyn_table t; build_table(minV, maxV, &t);
    for (int m = 0; m < 6; m++) for (int i = 0; i < 16; i++) lut[m * 16 + i] = t.lut[m][i];
    return t.base7Bit | (t.distance6Bit << 8);
}

void yko_curve_constants(float out[48]) {
    for (int i = 0; i < 16; i++) { float pos = i / 15.0f; out[i] = powf(pos, 1.4f); out[16 + i] = 1.0f - powf(1.0f - pos, 1.4f); }
    for (int i = 0; i < 8; i++)  { float pos = i / 7.0f;  out[32 + i] = powf(pos, 1.4f); out[40 + i] = 1.0f - powf(1.0f - pos, 1.4f); }
}

/* a10-a12: DynamicTileEncode (EncoderContext.cpp:4365-4602), full-resolution Y-type plane (isCo=isCg=isHalf*=false) */
int yko_dynamic_tile_encode(yko_enc* e, int mode3BitOnly, int plane, int32_t* dst) {
    check_mipmap_mask(e);
    const int w = e->w, h = e->h;
    const int32_t* src = e->plane[plane];
    int dw = w / 8, dh = h / 8;
    free(e->tileDefs); free(e->nibbles);
    e->tileDefs = (uint16_t*)malloc(sizeof(uint16_t) * ((size_t)dw * dh + 1));
    e->nibbles = (uint8_t*)calloc((size_t)dw * dh * 32 + 1, 1);
    e->nDefs = 0;
    int indexGlobal = 0;
    const int startMode = mode3BitOnly ? 3 : 0;

    /* constraint box (:4386-4391) and LeftRightOrder iteration incl. its quirks (encoder/framework.h:228-256) */
    int cx = (e->boundX0 >> 3) << 3, cy = (e->boundY0 >> 3) << 3;
    int cw = (((e->boundX1 + 7) >> 3) << 3) - cx, ch = (((e->boundY1 + 7) >> 3) << 3) - cy;
    int x = cx - 8, y = cy;
    for (;;) {
        int valid = (y < cy + ch);
        if (valid) {
            x += 8;
            if (x >= cx + cw) { x = cx; y += 8; valid = (y < h); }
        }
        int rw = (x + 8 > cw) ? (x % 8) : 8;
        int rh = (y + 8 > ch) ? (y % 8) : 8;
        if (!valid) break;

        /* Plane::GetMinMax_Y (Plane.cpp:489-587): valid = mask && !smooth */
        int mn = 99999999, mx = -99999999, any = 0;
        int maxY = y + rh > h ? h : y + rh, maxX = x + rw > w ? w : x + rw;
        for (int yy = y; yy < maxY; yy++) for (int xx = x; xx < maxX; xx++) {
            size_t i = xx + (size_t)yy * w;
            if (e->mipmapMask[i] && !(e->smoothMap && e->smoothMap[i])) {
                int V = src[i]; any = 1;
                if (V < mn) mn = V;
                if (V > mx) mx = V;
            }
        }
        if (!any) { mn = 0; mx = 0; }

        /* GetTileDynamic_Y (:747-1212) */
        int useSigned = 0;
        if (mn < 0) { mn += 128; mx += 128; useSigned = 1; }
        dyn_table tbl; build_table(mn, mx, &tbl);
        int bestMode = -1; float bestErr = 99999999.0f;
        int bestVal[64], bestCode[64], bVal[64], bCode[64];
        for (int mode = startMode; mode < 6; mode++) {
            int count = mode < 3 ? 16 : 8;
            const int* LUT = tbl.lut[mode];
            float errorDist = 0.0f;
            for (int n = 0; n < 64; n++) bVal[n] = -999;
            for (int ty = 0; ty < rh; ty++) for (int tx = 0; tx < rw; tx++) {
                size_t i = (size_t)(tx + x) + (size_t)(ty + y) * w;
                if (e->mipmapMask[i] && !(e->smoothMap && e->smoothMap[i])) {
                    int v = src[i] + (useSigned ? 128 : 0);
                    int minDiff = 99999, found = 0, valueFound = 0;
                    for (int n = 0; n < count; n++) {
                        int d = abs(LUT[n] - v);
                        if (d < minDiff) { minDiff = d; found = n; valueFound = LUT[n]; }
                    }
                    if (v != 0) errorDist += ((float)minDiff / (float)v);
                    bVal[tx + (ty << 3)] = valueFound; bCode[tx + (ty << 3)] = found;
                }
            }
            if (errorDist <= bestErr) {                                   /* later mode wins ties (:897) */
                bestErr = errorDist; bestMode = mode;
                memcpy(bestVal, bVal, sizeof bVal); memcpy(bestCode, bCode, sizeof bCode);
            }
        }
        int valueCount = 0;
        for (int ty = 0; ty < rh; ty++) for (int tx = 0; tx < rw; tx++) {  /* :1174-1190 */
            int v = bestVal[tx + (ty << 3)];
            if (v != -999) {
                e->nibbles[indexGlobal >> 1] |= (uint8_t)((indexGlobal & 1) ? (bestCode[tx + (ty << 3)] << 4) : bestCode[tx + (ty << 3)]);
                indexGlobal++; valueCount++;
                if (dst) dst[(x + tx) + (size_t)(y + ty) * w] = v;        /* :4448-4457 (offset 0 for Y planes) */
            }
        }
        if (valueCount) {
            /* TileInfo fields are u8 (:506-515); EncodeTileType (include/YAIK_private.h:358) then stored as u16 */
            unsigned type = (uint8_t)bestMode, range = (uint8_t)tbl.distance6Bit, base = (uint8_t)tbl.base7Bit;
            e->tileDefs[e->nDefs++] = (uint16_t)((type << 13) | (range << 7) | base);
        }
    }
    e->nNibbles = indexGlobal;
    return e->nDefs;
}

const uint16_t* yko_last_tile_defs(const yko_enc* e, int* n) { *n = e->nDefs; return e->tileDefs; }
const uint8_t* yko_last_nibbles(const yko_enc* e, int* nBytes, int* nNibbles) {
    *nNibbles = e->nNibbles; *nBytes = (e->nNibbles + 1) >> 1; return e->nibbles;   /* :4525-4527 close half byte */
}

/* ------------------------------------------------------------------------------------------- */
/* a15: DynamicTileCompressor (EncoderContext.cpp:8398-8522), colorCompression1D = 255, rangeCompression1D = 15 */
int yko_dynamic_tile_compressor(yko_enc* e, int plane, int32_t* debugOut) {
    const int w = e->w, h = e->h;
    const int32_t* src = e->plane[plane];
    const uint8_t* map = e->mapSmoothTile[plane];
    if (!map) return -1;
    if (!e->pix1d) {
        e->capPix1d = (size_t)w * h * 3 + 64; e->pix1d = (uint8_t*)malloc(e->capPix1d);
        e->capType1d = (size_t)(w / 8 + 1) * (h / 8 + 1) * 9 + 64; e->type1d = (uint8_t*)malloc(e->capType1d);
    }
    int tiles = 0;
    for (int y = 0; y < h; y += 8) for (int x = 0; x < w; x += 8) {
        uint8_t histo[256]; memset(histo, 0, sizeof histo);
        int values[64], offX[64], offY[64], pixelCount = 0;
        for (int y2 = 0; y2 < 8; y2 += 4) {
            int hasLeft  = map[clampi(x, 0, w - 1)     + (size_t)clampi(y2 + y, 0, h - 1) * w] == 0;
            int hasRight = map[clampi(x + 4, 0, w - 1) + (size_t)clampi(y2 + y, 0, h - 1) * w] == 0;
            if (hasLeft | hasRight) {
                int lengthX = (hasLeft && hasRight) ? 8 : 4;
                int x2 = (lengthX == 4 && hasRight) ? 4 : 0;
                for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < lengthX; ix++) {
                    int v = yko_compress_f(px(src, w, h, x + x2 + ix, y + y2 + iy), 255);
                    histo[v & 255]++;
                    values[pixelCount] = v; offX[pixelCount] = x2 + ix; offY[pixelCount] = y2 + iy; pixelCount++;
                }
            }
        }
        if (pixelCount <= 0) continue;
        /* FindAndRemoveMostUsedColor (:8335-8356): right-most maximum, clamped to 1..254, +-1 removed */
        int best = -1, bestV = -1;
        for (int n = 0; n < 256; n++) if (histo[n] >= bestV) { bestV = histo[n]; best = n; }
        if (best == 0) best = 1;
        if (best == 255) best = 254;
        histo[best - 1] = 0; histo[best] = 0; histo[best + 1] = 0;
        /* Model1 (:8358-8381) */
        int minV = 99999, maxV = -99999;
        for (int n = 0; n < 256; n++) if (histo[n]) { if (minV > n) minV = n; if (maxV < n) maxV = n; }
        int minCol = 0, delta = 0;
        if (minV != 99999) { minCol = minV; delta = maxV - minV; }
        for (int n = 0; n < pixelCount; n++) {
            int v = values[n], outV;
            if (v >= best - 1 && v <= best + 1) { e->pix1d[e->nPix1d++] = 0; outV = best; }
            else {
                int idx = delta ? (((v - minCol) * 15) + ((delta >> 1) - 1)) / delta : 0;   /* GetValueModel1 (:8383) */
                e->pix1d[e->nPix1d++] = (uint8_t)(1 + idx);
                outV = minCol + ((idx * delta) / 15);                                       /* DecompModel1 (:8393) */
            }
            if (debugOut) debugOut[(x + offX[n]) + (size_t)(y + offY[n]) * w] = outV;
        }
        e->type1d[e->nType1d++] = (uint8_t)best;
        e->type1d[e->nType1d++] = (uint8_t)minCol;
        e->type1d[e->nType1d++] = (uint8_t)delta;
        tiles++;
    }
    return tiles;
}
const uint8_t* yko_1d_pix_stream(const yko_enc* e, int* n) { *n = e->nPix1d; return e->pix1d; }
const uint8_t* yko_1d_type_stream(const yko_enc* e, int* n) { *n = e->nType1d; return e->type1d; }

/* ------------------------------------------------------------------------------------------- */
/* §8(f)1: PaletteCompressor (EncoderContext.cpp:3259-3502) incl. registerCodeBook (:3231), FindCodeBook (:3248),
 * compCode (:3222).  Delta / code-book codec of the corner-colour stream.  Reproduces the reference's quirk that
 * FindCodeBook scans rows 0..63 of a table whose stale rows survive from earlier calls.  glibc's qsort is a stable
 * merge sort for these sizes, so equal ref counts keep registration order. */
static void code_register(yko_enc* e, int dr, int dg, int db) {
    for (int n = 0; n < e->codeCount; n++)
        if (e->codeRGB[n][1] == dr && e->codeRGB[n][2] == dg && e->codeRGB[n][3] == db) { e->codeRGB[n][0]++; return; }
    int* c = e->codeRGB[e->codeCount++];
    c[0] = 0; c[1] = dr; c[2] = dg; c[3] = db;
}
static int code_find(const yko_enc* e, int dr, int dg, int db) {
    for (int n = 0; n < 64; n++)
        if (e->codeRGB[n][1] == dr && e->codeRGB[n][2] == dg && e->codeRGB[n][3] == db) return n;
    return -1;
}

int yko_palette_compress(yko_enc* e, const uint8_t* input, int size) {
    int entryCol = size / 3;
    if (!e->codeRGB) { e->codeCap = 100000; e->codeRGB = (int (*)[4])calloc((size_t)e->codeCap, sizeof(int[4])); }
    if (entryCol + 2 > e->codeCap) return -1;
    int lmax = size * 3, si = 0;
    free(e->lastPalette);
    uint8_t* out = e->lastPalette = (uint8_t*)malloc((size_t)lmax + 16);
    e->lastPaletteBytes = 0;
    uint8_t* decomp = (uint8_t*)malloc((size_t)size + 16);
    uint8_t* ds = decomp;
    int ok = 1;
#define WR(v) do { if (si < lmax) out[si++] = (uint8_t)(v); else { ok = 0; goto done; } } while (0)
    e->codeCount = 0;
    code_register(e, 0, 0, 0);
    for (int n = 1; n < entryCol; n++) {                                   /* phase 1 (:3288-3309) */
        const uint8_t* pix = &input[n * 3];
        int prevStart = n - 64; if (prevStart < 0) prevStart = 0;
        int distMin = 999999999, bR = 0, bG = 0, bB = 0;
        for (int prev = prevStart; prev < n; prev++) {
            int dR = pix[0] - input[prev * 3], dG = pix[1] - input[prev * 3 + 1], dB = pix[2] - input[prev * 3 + 2];
            int ld = dR * dR + dG * dG + dB * dB;
            if (ld < distMin) { distMin = ld; bR = dR; bG = dG; bB = dB; }
        }
        code_register(e, bR, bG, bB);
    }
    /* stable sort rows 1..count-1 by ref descending (:3315) */
    for (int i = 2; i < e->codeCount; i++) {
        int t[4]; memcpy(t, e->codeRGB[i], sizeof t);
        int j = i - 1;
        while (j >= 1 && e->codeRGB[j][0] < t[0]) { memcpy(e->codeRGB[j + 1], e->codeRGB[j], sizeof t); j--; }
        memcpy(e->codeRGB[j + 1], t, sizeof t);
    }
    int finalCount = e->codeCount > 128 ? 128 : e->codeCount;
    WR(finalCount);
    for (int n = 0; n < finalCount; n++) { WR(e->codeRGB[n][1]); WR(e->codeRGB[n][2]); WR(e->codeRGB[n][3]); }
    WR(input[0]); WR(input[1]); WR(input[2]);
    *ds++ = input[0]; *ds++ = input[1]; *ds++ = input[2];
    const uint8_t* prevCol = decomp;
    for (int n = 1; n < entryCol; n++) {                                   /* phase 2 (:3345-3487) */
        const uint8_t* pix = &input[n * 3];
        int prevStart = n - 65; if (prevStart < 0) prevStart = 0;
        int done = 0, bestIdx = 999, bestDist = 0;
        for (int prev = n - 1; prev >= prevStart; prev--) {
            int dR = pix[0] - input[prev * 3], dG = pix[1] - input[prev * 3 + 1], dB = pix[2] - input[prev * 3 + 2];
            int index = code_find(e, dR, dG, dB);
            if (index >= 0) {
                if (prev == n - 1) {
                    WR(index & 0x7F); done = 1;
                    ds[0] = (uint8_t)(prevCol[0] + dR); ds[1] = (uint8_t)(prevCol[1] + dG); ds[2] = (uint8_t)(prevCol[2] + dB);
                    ds += 3; prevCol = ds - 3;
                    break;
                } else {
                    int distance = (n - prev) - 2;
                    if (distance < 64 && index < bestIdx) { bestIdx = index; bestDist = distance; done = 1; }
                }
            }
        }
        if (bestIdx != 999) {
            WR(0xC0 | (bestDist & 0x3F));
            prevCol = ds - (bestDist + 2) * 3;
            WR(bestIdx & 0x7F); done = 1;
            ds[0] = (uint8_t)(prevCol[0] + e->codeRGB[bestIdx][1]); ds[1] = (uint8_t)(prevCol[1] + e->codeRGB[bestIdx][2]);
            ds[2] = (uint8_t)(prevCol[2] + e->codeRGB[bestIdx][3]);
            ds += 3; prevCol = ds - 3;
        }
        if (!done) {
            int dR = pix[0] - pix[-3], dG = pix[1] - pix[-2], dB = pix[2] - pix[-1];
            int mask = (dR ? 1 : 0) | (dG ? 2 : 0) | (dB ? 4 : 0);
            if (dR >= -128 && dR <= 127 && dG >= -128 && dG <= 127 && dB >= -128 && dB <= 127) {
                WR(0x80 | mask);
                if (dR) WR(dR);
                if (dG) WR(dG);
                if (dB) WR(dB);
                ds[0] = (uint8_t)(prevCol[0] + dR); ds[1] = (uint8_t)(prevCol[1] + dG); ds[2] = (uint8_t)(prevCol[2] + dB);
                ds += 3; prevCol = ds - 3;
            } else {
                WR(0x88 | mask);
                ds[0] = dR ? pix[0] : prevCol[0]; ds[1] = dG ? pix[1] : prevCol[1]; ds[2] = dB ? pix[2] : prevCol[2];
                ds += 3; prevCol = ds - 3;
                if (dR) WR(pix[0]);
                if (dG) WR(pix[1]);
                if (dB) WR(pix[2]);
            }
        }
    }
#undef WR
done:
    free(decomp);
    e->lastPaletteBytes = ok ? si : 0;
    return ok ? si : -1;
}
const uint8_t* yko_last_palette(const yko_enc* e, int* n) { *n = e->lastPaletteBytes; return e->lastPalette; }

/* PaletteDecompressor (decoder/YAIK_GenericFunctions.cpp:139-241): inverse of the above + PaletteFullRangeRemapping.
 * input must be readable for inputSize + 384 bytes (the decoder's "secure buffer"). Returns 1 on success. */
int yko_palette_decompress(const uint8_t* input, int inputSize, uint8_t* output, int outputSize, int colorCompression) {
    const uint8_t* in = input;
    int codeBookSize = *in++;
    int pos = 1 + codeBookSize * 3;
    if (pos > inputSize) return 0;
    const uint8_t* codeBook = in; in += codeBookSize * 3;
    const uint8_t* inEnd = input + 1 + inputSize + 128 * 3;
    uint8_t* wr = output; uint8_t* lastRGB = output + outputSize - 3;
    *wr++ = *in++; *wr++ = *in++; *wr++ = *in++;
    const uint8_t* last = output;
    while (wr <= lastRGB) {
        if (in >= inEnd) return 0;
        int c = *in++;
        if (c & 0x80) {
            if (c & 0x40) {
                last = wr - ((c & 0x3F) + 2) * 3;
                if (last < output) return 0;
            } else {
                switch ((c >> 3) & 7) {
                case 0:
                    wr[0] = (uint8_t)(last[0] + ((c & 1) ? *in++ : 0)); wr[1] = (uint8_t)(last[1] + ((c & 2) ? *in++ : 0));
                    wr[2] = (uint8_t)(last[2] + ((c & 4) ? *in++ : 0)); break;
                case 1:
                    wr[0] = (c & 1) ? *in++ : last[0]; wr[1] = (c & 2) ? *in++ : last[1]; wr[2] = (c & 4) ? *in++ : last[2]; break;
                default: return 0;
                }
                last = wr; wr += 3;
            }
        } else {
            const uint8_t* code = &codeBook[(c & 0x7F) * 3];
            wr[0] = (uint8_t)(last[0] + code[0]); wr[1] = (uint8_t)(last[1] + code[1]); wr[2] = (uint8_t)(last[2] + code[2]);
            last = wr; wr += 3;
        }
    }
    yko_palette_remap(output, outputSize, colorCompression);
    return 1;
}

/* ------------------------------------------------------------------------------------------- */
/* decoder side: YAIK_Instance buffers as allocated by YAIK_DecodeImage (decoder/YAIK_API.cpp:650-657, 855-874) */
yko_dec* yko_dec_create(int w, int h) {
    yko_dec* d = (yko_dec*)calloc(1, sizeof *d);
    d->w = w; d->h = h; d->tileW = (w + 7) >> 3; d->tileH = (h + 7) >> 3;
    d->planeSize = d->tileW * d->tileH * 64;
    d->planes = (uint8_t*)calloc((size_t)d->planeSize * 3, 1);
    d->strideRGBMap = (w >> 2) + 1;
    d->lattice = d->strideRGBMap * ((h >> 2) + 1);
    d->mapRGB = (uint8_t*)calloc((size_t)d->lattice * 3, 1);
    d->sizeMapMask = (d->lattice + 7) >> 3;
    d->mapRGBMask = (uint8_t*)calloc((size_t)d->sizeMapMask * 3, 1);
    d->stride4 = (w + 15) >> 4;
    d->tile4x4MaskSize = ((d->stride4 << 2) * (((h + 7) >> 3) << 1)) >> 3;
    d->tile4x4Mask = (uint8_t*)calloc((size_t)d->tile4x4MaskSize * 3, 1);
    d->singleRGB = 1;
    return d;
}
void yko_dec_destroy(yko_dec* d) {
    if (!d) return;
    free(d->planes); free(d->mapRGB); free(d->mapRGBMask); free(d->tile4x4Mask); free(d);
}

/* a16: the seven DecompressGradientWxH loops share one structure (decoder/YAIK_Gradient.cpp:28-1418):
 * walk swizzle blocks row-major, tiles row-major inside a block; for a set bit pull the not-yet-seen
 * lattice corners TL,TR,BL,BR from the stream, then integer bilinear with truncation, and flag the
 * covered 4x4 cells in tile4x4Mask (bit = x8*4 + y4*2 + x4 of a 16x8 area, e.g. :951-953). */
int yko_dec_gradient(yko_dec* d, int sx, int sy, const uint8_t* bitmap, int bitmapBytes, const uint8_t* rgb, int rgbBytes) {
    int bigX, bigY, bitCount;
    swizzle_size(sx, sy, &bigX, &bigY, &bitCount);
    if (!bigX) return -1;
    const int w = d->w, h = d->h, TX = 1 << sx, TY = 1 << sy;
    int xBB = (w + bigX - 1) / bigX, yBB = (h + bigY - 1) / bigY;
    if (((xBB * yBB * bitCount) >> 3) > bitmapBytes) return -2;
    int rd = 0;
    const int tilesPerRow = bigX / TX;
    for (int by = 0; by < yBB; by++) for (int bx = 0; bx < xBB; bx++) {
        int posBlock = (by * xBB + bx) * bitCount;
        for (int t = 0; t < bitCount; t++) {
            int pos = posBlock + t;
            if (!(bitmap[pos >> 3] & (1 << (pos & 7)))) continue;
            int x = bx * bigX + (t % tilesPerRow) * TX, y = by * bigY + (t / tilesPerRow) * TY;
            if (x + TX > w || y + TY > h) continue;
            int li[4];
            li[0] = (x >> 2) + (y >> 2) * d->strideRGBMap;  li[1] = li[0] + (TX >> 2);
            li[2] = li[0] + (TY >> 2) * d->strideRGBMap;    li[3] = li[2] + (TX >> 2);
            for (int k = 0; k < 4; k++) {
                if (!(d->mapRGBMask[li[k] >> 3] & (1 << (li[k] & 7)))) {
                    d->mapRGBMask[li[k] >> 3] |= (uint8_t)(1 << (li[k] & 7));
                    for (int c = 0; c < 3; c++) d->mapRGB[li[k] * 3 + c] = (rd < rgbBytes) ? rgb[rd] : 0, rd++;
                }
            }
            for (int c = 0; c < 3; c++) {
                int TL = d->mapRGB[li[0] * 3 + c], TR = d->mapRGB[li[1] * 3 + c];
                int BL = d->mapRGB[li[2] * 3 + c], BR = d->mapRGB[li[3] * 3 + c];
                uint8_t* pl = d->planes + (size_t)c * d->planeSize;
                for (int ty = 0; ty < TY; ty++) {
                    int L = TL * (TY - ty) + BL * ty, R = TR * (TY - ty) + BR * ty;
                    for (int tx = 0; tx < TX; tx++) {
                        int xx = x + tx, yy = y + ty;
                        pl[(((yy >> 3) * d->tileW) + (xx >> 3)) * 64 + (yy & 7) * 8 + (xx & 7)] =
                            (uint8_t)((L * (TX - tx) + R * tx) >> (sx + sy));
                    }
                }
            }
            for (int cy = y >> 2; cy < (y + TY) >> 2; cy++) for (int cxx = x >> 2; cxx < (x + TX) >> 2; cxx++)
                d->tile4x4Mask[(cxx >> 2) + (cy >> 1) * d->stride4] |= (uint8_t)(1 << ((((cxx >> 1) & 1) << 2) | ((cy & 1) << 1) | (cxx & 1)));
        }
    }
    return rd;
}

/* a16, partial planes: DecompressGradient4x4R / G / B / RG / GB / RB (decoder/YAIK_Gradient.cpp:1208-1226 dispatch, bodies :1420-2732).
 * Same walk as the RGB 4x4 loop; per set bit and per corner TL,TR,BL,BR one byte is popped for every PRESENT plane whose own
 * mapRGBMask plane does not have the lattice point yet (e.g. :1516-1580), only the present planes are filled and only their
 * tile4x4Mask planes are marked (:1609-1612; with the reference's defects unless consistentMarks, see below).  The masks must be
 * in per-plane form (UpdateTileAndRGBMask, YAIK_API.cpp:875-877).
 * planeBit 7 = yko_dec_gradient.  Only the 4x4 size has partial-plane decoders in the reference. */
int yko_dec_gradient_planes(yko_dec* d, int planeBit, int consistentMarks, const uint8_t* bitmap, int bitmapBytes, const uint8_t* rgb, int rgbBytes) {
    if (planeBit == 7) return yko_dec_gradient(d, 2, 2, bitmap, bitmapBytes, rgb, rgbBytes);
    if (planeBit < 1 || planeBit > 6) return -1;
    const int w = d->w, h = d->h, bigX = 32, bigY = 32, bitCount = 64, TX = 4, TY = 4;
    int xBB = (w + bigX - 1) / bigX, yBB = (h + bigY - 1) / bigY;
    if (((xBB * yBB * bitCount) >> 3) > bitmapBytes) return -2;
    int rd = 0;
    for (int by = 0; by < yBB; by++) for (int bx = 0; bx < xBB; bx++) {
        int posBlock = (by * xBB + bx) * bitCount;
        for (int t = 0; t < bitCount; t++) {
            int pos = posBlock + t;
            if (!(bitmap[pos >> 3] & (1 << (pos & 7)))) continue;
            int x = bx * bigX + (t % 8) * TX, y = by * bigY + (t / 8) * TY;
            if (x >= w || y >= h) continue;
            int li[4];
            li[0] = (x >> 2) + (y >> 2) * d->strideRGBMap;  li[1] = li[0] + 1;
            li[2] = li[0] + d->strideRGBMap;                li[3] = li[2] + 1;
            for (int k = 0; k < 4; k++) for (int c = 0; c < 3; c++) {
                if (!(planeBit & (1 << c))) continue;
                uint8_t* has = d->mapRGBMask + (size_t)d->sizeMapMask * c;
                if (!(has[li[k] >> 3] & (1 << (li[k] & 7)))) {
                    has[li[k] >> 3] |= (uint8_t)(1 << (li[k] & 7));
                    d->mapRGB[li[k] * 3 + c] = (rd < rgbBytes) ? rgb[rd] : 0; rd++;
                }
            }
            for (int c = 0; c < 3; c++) {
                if (!(planeBit & (1 << c))) continue;
                int TL = d->mapRGB[li[0] * 3 + c], TR = d->mapRGB[li[1] * 3 + c];
                int BL = d->mapRGB[li[2] * 3 + c], BR = d->mapRGB[li[3] * 3 + c];
                uint8_t* pl = d->planes + (size_t)c * d->planeSize;
                for (int ty = 0; ty < TY; ty++) {
                    int L = TL * (TY - ty) + BL * ty, R = TR * (TY - ty) + BR * ty;
                    for (int tx = 0; tx < TX; tx++) {
                        int xx = x + tx, yy = y + ty;
                        pl[(((yy >> 3) * d->tileW) + (xx >> 3)) * 64 + (yy & 7) * 8 + (xx & 7)] = (uint8_t)((L * (TX - tx) + R * tx) >> 4);
                    }
                }
                /* tile4x4Mask marking.  consistentMarks: the plane's own mask, which is what the encoder's per-plane coverage and
                 * Decompress1D assume.  Otherwise AS THE REFERENCE DOES IT: only the two-plane variants mark at all (the R / G / B loops
                 * :2138-2732 never touch tile4x4Mask), and GB / RB put the B marks at tile4x4Mask + (tile4x4MaskSize >> 1), i.e. inside
                 * plane 0 of the mask, instead of + (tile4x4MaskSize << 1) (:1678, :1924). */
                int cxx = x >> 2, cy = y >> 2;
                size_t base;
                if (consistentMarks) base = (size_t)d->tile4x4MaskSize * c;
                else {
                    if (planeBit == 1 || planeBit == 2 || planeBit == 4) continue;
                    base = (c == 2) ? (size_t)(d->tile4x4MaskSize >> 1) : (size_t)d->tile4x4MaskSize * c;
                }
                d->tile4x4Mask[base + (cxx >> 2) + (cy >> 1) * d->stride4] |= (uint8_t)(1 << ((((cxx >> 1) & 1) << 2) | ((cy & 1) << 1) | (cxx & 1)));
            }
        }
    }
    return rd;
}

void yko_dec_split_masks(yko_dec* d) {
    if (!d->singleRGB) return;                                             /* once (YAIK_API.cpp:533-534) */
    d->singleRGB = 0;
    for (int p = 1; p < 3; p++) {
        memcpy(d->mapRGBMask + (size_t)d->sizeMapMask * p, d->mapRGBMask, (size_t)d->sizeMapMask);
        memcpy(d->tile4x4Mask + (size_t)d->tile4x4MaskSize * p, d->tile4x4Mask, (size_t)d->tile4x4MaskSize);
    }
}

/* a17: Decompress1D (decoder/YAIK_3DTile.cpp:24-240) */
int yko_dec_1d(yko_dec* d, int plane, const uint8_t* type, int* typePos, const uint8_t* pix, int* pixPos, int compressionRange) {
    if (plane < 0 || plane > 2) return -1;
    const int w = d->w, h = d->h;
    const uint8_t* used = d->tile4x4Mask + (size_t)d->tile4x4MaskSize * plane;
    uint8_t* pl = d->planes + (size_t)d->planeSize * plane;
    int inv = (1 << 24) / compressionRange, tiles = 0;
    int tp = *typePos, pp = *pixPos;
    for (int y = 0; y < h; y += 8) for (int x = 0; x < w; x += 8) {
        int q = used[(x >> 4) + (y >> 3) * d->stride4];
        q = (q >> ((x & 8) ? 4 : 0)) & 0xF;
        if (q == 0xF) continue;
        int color0 = type[tp++], base = type[tp++], delta = type[tp++];
        int delta2 = ((delta * inv) >> 8) + 1;
        uint8_t* tile = pl + (((y >> 3) * d->tileW) + (x >> 3)) * 64;
        tiles++;
        for (int half = 0; half < 2; half++) {
            int left = !(q & 1), right = !(q & 2);
            q >>= 2;
            for (int r = 0; r < 4; r++) {
                for (int cx = 0; cx < 8; cx++) {
                    if ((cx < 4 && !left) || (cx >= 4 && !right)) continue;
                    int L = pix[pp++];
                    tile[(half * 4 + r) * 8 + cx] = (uint8_t)(L ? (base + (((L - 1) * delta2) >> 16)) : color0);
                }
            }
        }
    }
    *typePos = tp; *pixPos = pp;
    return tiles;
}

/* a18: Decompress1BitTiled (decoder/YAIK_Mipmap.cpp:23-154), mipmapLevel 4 (the only implemented one, :139-147): source bit
 * `pos` (tile-bbox row-major, LSB first, :124) becomes a 16x16 block of ones in a mask stored as u64 words, 16 pixels x 4 rows
 * per word pair: per tile the reference writes v,v into row group A and v,v into row group B = A + 2*tileWidth words, then both
 * cursors skip the other group (:119-136).  Pinned against the compiled reference (blob dec_mask of oracle/ref_driver.cpp). */
int yko_dec_mask(const uint8_t* bits, int bw, int bh, uint8_t* out) {
    uint64_t* tile = (uint64_t*)out;
    uint64_t* A = tile; uint64_t* B = A + ((size_t)bw << 1);
    int pos = 0;
    for (int y = 0; y < bh; y++) {
        for (int x = 0; x < bw; x++) {
            uint64_t v = (bits[pos >> 3] & (1 << (pos & 7))) ? ~(uint64_t)0 : 0;
            *A++ = v; *A++ = v; *B++ = v; *B++ = v;
            pos++;
        }
        A += (size_t)bw << 1; B += (size_t)bw << 1;
    }
    return (bw * bh * 256) >> 3;
}

/* a20: internal_imageBuilderFunc (decoder/YAIK_DefaultCallback.cpp:24-191) for widths / heights that are multiples of 8 (the
 * clipped right / bottom branches :85-190 are not restated: Image::LoadPNG only admits multiples of 8, Image.cpp:206).
 * planes = R|G|B 8x8-tiled u8 (planeSize bytes each); out rows are `stride` bytes apart, bytes the loop does not write are left.
 *   alpha == NULL: 3 bytes per pixel (:64-79).
 *   alpha != NULL: the reference's RGBA branch AS IT IS (:45-62): `*dst = *pAlpha++` stores the alpha byte WITHOUT advancing dst,
 *   so the next pixel's red overwrites it -> rows of RGB triples followed by ONE alpha byte at offset 3*w; and the alpha row
 *   cursors pAL[n] (:36-39) are never moved to the next tile row (only dst_pL is, :128-130), so tile row t reads alpha rows
 *   t .. t+7.  Reproduced, not fixed: the product documents its own RGBA layout separately (include/yaik_hip.h). */
int yko_image_builder(const uint8_t* planes, int planeSize, int w, int h, const uint8_t* alpha, int strideA, uint8_t* out, int stride) {
    if ((w & 7) || (h & 7)) return -1;
    const uint8_t* pR = planes; const uint8_t* pG = planes + planeSize; const uint8_t* pB = planes + 2 * (size_t)planeSize;
    uint8_t* rowL[8]; const uint8_t* pAL[8];
    for (int n = 0; n < 8; n++) { rowL[n] = out + (size_t)stride * n; pAL[n] = alpha ? alpha + (size_t)strideA * n : NULL; }
    for (int ty = 0; ty < h; ty += 8) {
        uint8_t* dst[8];
        for (int n = 0; n < 8; n++) dst[n] = rowL[n];
        for (int tx = 0; tx < (w >> 3); tx++) {
            for (int n = 0; n < 8; n++) {
                uint8_t* d = dst[n];
                for (int k = 0; k < 8; k++) {
                    *d++ = *pR++; *d++ = *pG++; *d++ = *pB++;
                    if (alpha) *d = *pAL[n]++;
                }
                dst[n] = d;
            }
        }
        for (int n = 0; n < 8; n++) rowL[n] += (size_t)stride << 3;
    }
    return 0;
}

const uint8_t* yko_dec_planes(const yko_dec* d, int* planeSize) { *planeSize = d->planeSize; return d->planes; }
const uint8_t* yko_dec_tile4x4(const yko_dec* d, int* n) { *n = d->tile4x4MaskSize; return d->tile4x4Mask; }
const uint8_t* yko_dec_map_rgb(const yko_dec* d, int* n) { *n = d->lattice * 3; return d->mapRGB; }
const uint8_t* yko_dec_map_rgb_mask(const yko_dec* d, int* n) { *n = d->sizeMapMask; return d->mapRGBMask; }
