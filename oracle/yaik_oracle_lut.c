/* TEST INFRASTRUCTURE ONLY — see yaik_oracle.h.  Plain-C restatement of the reference's 3-D LUT tile search, SURVEY 8(f)4:
 *   EncoderContext::Load3DPattern            encoder/EncoderContext.cpp:7851-7917   (sortPalette :2920-2960, morton tables :2799-2908)
 *   EvalCtx3D::Set3DPointCloud               :4744-4814
 *   EncoderContext::Correlation3DSearch      :6245-6781   (buildBBox3D :132-193, EvalCtx3D::EvaluatePoint3D / GetEvaluation3D
 *                                                          encoder/EncoderContext.h:629-711, swap3D :5314-5354, swap3DTable :5356-5390)
 *   EncoderContext::computeValues3D          :5807-6094
 *   EndCorrelationSearch's stream finishing  :7494-7497 (CompressF on the colour stream), :7526-7529 (indices x 3)
 * Written from the behaviour of the reference; no reference code is copied.  Float expressions keep the reference's order
 * (build with -ffp-contract=off).  The reference's own LUT bank is not in its repository: tests feed a synthetic one (tests/lutbank.py).
 */
#include "yaik_oracle_internal.h"
#include <stdlib.h>
#include <string.h>

#define LUT_FACTOR 128                       /* FACTOR, EncoderContext.cpp:22 */
#define CUBE (64 * 64 * 64)

typedef struct {
    int count;
    uint8_t pts[64 * 3];                     /* after the morton sort */
    int16_t fac[4][3][64];                   /* [6,5,4,3 bit][x,y,z][entry] = xFactorNBit ... */
    int32_t* dist;                           /* distanceField3D */
    uint8_t* pos[4];                         /* position6Bit3D .. position3Bit3D */
    uint8_t value[4][128];                   /* value6Bit .. value3Bit of the last computeValues3D on this pattern */
    int sum[48], samples;
} lut_pattern;

struct yko_lut_state {
    lut_pattern pat[64]; int nPat;
    /* StartCorrelationSearch (:7316-7364) */
    uint16_t* tileType; int nType;
    uint8_t* color; int nColor;
    uint8_t* idx[4]; int nIdx[4];            /* [0] = 3 bit .. [3] = 6 bit */
    uint8_t* map[6]; int mapBytes[6];        /* 16x8, 8x16, 8x8, 8x4, 4x8, 4x4 */
    int32_t* preview[3];
};

void yko_lut_free(struct yko_lut_state* s) {
    if (!s) return;
    for (int e = 0; e < 64; e++) { free(s->pat[e].dist); for (int k = 0; k < 4; k++) free(s->pat[e].pos[k]); }
    free(s->tileType); free(s->color);
    for (int k = 0; k < 4; k++) free(s->idx[k]);
    for (int k = 0; k < 6; k++) free(s->map[k]);
    for (int k = 0; k < 3; k++) free(s->preview[k]);
    free(s);
}

/* bit k of v -> bit 3k + axis (morton256_x/y/z, :2799-2908) */
static uint32_t morton3(int r, int g, int b) {
    uint32_t m = 0;
    for (int k = 0; k < 8; k++) m |= (uint32_t)((r >> k) & 1) << (3 * k) | (uint32_t)((g >> k) & 1) << (3 * k + 1) | (uint32_t)((b >> k) & 1) << (3 * k + 2);
    return m;
}

/* Load3DPattern: interleave, selection sort by morton code (sortPalette: the minimum of the rest is swapped to the front, first minimum
 * wins), Set3DPointCloud.  count <= 64 (beyond that the reference reads past its reduced array, :7907-7917). */
int yko_lut_load(yko_enc* e, const uint8_t* r, const uint8_t* g, const uint8_t* b, int count) {
    if (count < 1 || count > 64) return -1;
    if (!e->lut) e->lut = (struct yko_lut_state*)calloc(1, sizeof(struct yko_lut_state));
    struct yko_lut_state* S = e->lut;
    if (S->nPat >= 64) return -1;                                              /* "LUT 3D more than 64 entries" :7912 */
    lut_pattern* P = &S->pat[S->nPat];
    P->count = count;
    for (int n = 0; n < count; n++) { P->pts[n * 3] = r[n]; P->pts[n * 3 + 1] = g[n]; P->pts[n * 3 + 2] = b[n]; }
    for (int i = 0; i < count - 1; i++) {
        int mn = i;
        for (int j = i + 1; j < count; j++)
            if (morton3(P->pts[mn * 3], P->pts[mn * 3 + 1], P->pts[mn * 3 + 2]) > morton3(P->pts[j * 3], P->pts[j * 3 + 1], P->pts[j * 3 + 2])) mn = j;
        if (mn != i) for (int c = 0; c < 3; c++) { uint8_t t = P->pts[i * 3 + c]; P->pts[i * 3 + c] = P->pts[mn * 3 + c]; P->pts[mn * 3 + c] = t; }
    }
    /* Set3DPointCloud: factor tables (every 1st / 2nd / 4th / 8th point), (p / 63.0f) * FACTOR truncated to s16 */
    memset(P->fac, 0, sizeof P->fac);
    for (int step = 0; step < 4; step++)
        for (int pts = 0; pts < count; pts += 1 << step)
            for (int c = 0; c < 3; c++) P->fac[step][c][pts >> step] = (int16_t)((P->pts[pts * 3 + c] / 63.0f) * LUT_FACTOR);
    P->dist = (int32_t*)malloc(sizeof(int32_t) * CUBE);
    for (int k = 0; k < 4; k++) P->pos[k] = (uint8_t*)malloc(CUBE);
    /* nearest point of every cell, first minimum wins; distanceField3D is overwritten by every step, so it ends as the distance to the
     * nearest of the 3-bit subset (every 8th point) */
    for (int step = 0; step < 4; step++)
        for (int z = 0; z < 64; z++) for (int y = 0; y < 64; y++) for (int x = 0; x < 64; x++) {
            int minDist = 999999999;
            const int i3 = x + (y << 6) + (z << 12);
            for (int pts = 0; pts < count; pts += 1 << step) {
                const int dx = x - P->pts[pts * 3], dy = y - P->pts[pts * 3 + 1], dz = z - P->pts[pts * 3 + 2];
                const int d = dx * dx + dy * dy + dz * dz;
                if (d < minDist) { minDist = d; P->dist[i3] = d; P->pos[step][i3] = (uint8_t)(pts >> step); }
            }
        }
    return S->nPat++;
}

int yko_lut_count(const yko_enc* e) { return e->lut ? e->lut->nPat : 0; }
const int16_t* yko_lut_factors(const yko_enc* e, int pattern) { return &e->lut->pat[pattern].fac[0][0][0]; }
const int32_t* yko_lut_distance_field(const yko_enc* e, int pattern) { return e->lut->pat[pattern].dist; }
const uint8_t* yko_lut_positions(const yko_enc* e, int pattern, int step) { return e->lut->pat[pattern].pos[step]; }

static void swizzle_size_lut(int sx, int sy, int* bigX, int* bigY, int* bitCount) {       /* getSwizzleSize, include/YAIK_private.h:212-276 */
    *bigX = sx == 2 ? 32 : 64; *bigY = sy == 2 ? 32 : 64; *bitCount = (*bigX >> sx) * (*bigY >> sy);
}

/* StartCorrelationSearch (:7316-7364) */
void yko_lut_start(yko_enc* e) {
    if (!e->lut) e->lut = (struct yko_lut_state*)calloc(1, sizeof(struct yko_lut_state));
    struct yko_lut_state* S = e->lut;
    const int w = e->w, h = e->h;
    static const int sz[6][2] = { {4,3}, {3,4}, {3,3}, {3,2}, {2,3}, {2,2} };
    free(S->tileType); free(S->color);
    for (int k = 0; k < 4; k++) { free(S->idx[k]); S->idx[k] = (uint8_t*)calloc((size_t)w * h + 128, 1); S->nIdx[k] = 0; }
    for (int k = 0; k < 6; k++) {
        int bx, by, bc; swizzle_size_lut(sz[k][0], sz[k][1], &bx, &by, &bc);
        S->mapBytes[k] = ((w + bx - 1) / bx) * ((h + by - 1) / by) * bc;        /* BitmapSwizzleMapSize :7310 (bits, used as a byte count) */
        free(S->map[k]); S->map[k] = (uint8_t*)calloc((size_t)S->mapBytes[k] + 1, 1);
    }
    const size_t maxTiles = (size_t)(w / 4) * ((size_t)h * 4) + 16;
    S->tileType = (uint16_t*)calloc(maxTiles, 2); S->nType = 0;
    S->color = (uint8_t*)calloc(maxTiles * 6, 1); S->nColor = 0;
    for (int k = 0; k < 3; k++) { free(S->preview[k]); S->preview[k] = (int32_t*)calloc((size_t)w * h, sizeof(int32_t)); }
    if (!e->mapSmoothTile[0])
        for (int n = 0; n < 3; n++) e->mapSmoothTile[n] = (uint8_t*)calloc((size_t)w * h, 1);
}

static void swap3(int mode, int* x, int* y, int* z) {                           /* swap3D :5314-5354 */
    int t;
    switch (mode) {
    case 1: t = *z; *z = *y; *y = t; break;
    case 2: t = *x; *x = *y; *y = t; break;
    case 3: t = *x; *x = *y; *y = *z; *z = t; break;
    case 4: t = *y; *y = *x; *x = *z; *z = t; break;
    case 5: t = *x; *x = *z; *z = t; break;
    default: break;
    }
}

enum { M3 = 0, M4 = 1, M5 = 2, M6 = 3, SKIP = 4 };                            /* EncoderContext::Mode, EncoderContext.h:377-385 */

/* computeValues3D (:5807-6094).  tile[4][128*3]: decoded colours per bit depth ([0] = 6 bit .. [3] = 3 bit), only entries of coded pixels. */
static int compute_values(yko_enc* e, lut_pattern* P, int tileSizeX, int tileSizeY, const uint8_t* mask, int mode, int px, int py,
                          const int bb[6] /* x0,y0,z0,x1,y1,z1 */, int* minDiff, int (*tile)[128 * 3]) {
    const int w = e->w;
    int res = SKIP, reject = 0, streamIdx = 0;
    int absErr[4] = { 0, 0, 0, 0 }, wrong[4] = { 0, 0, 0, 0 };                 /* [0] = 6 bit .. [3] = 3 bit */
    const int d[3] = { bb[3] - bb[0], bb[4] - bb[1], bb[5] - bb[2] };
    int xCount = 1;
    if (tileSizeX > 8) { tileSizeX = 8; xCount = 2; }
    for (int xa = 0; xa < xCount; xa++) {
        for (int y = 0; y < tileSizeY; y++) {
            for (int x = 0; x < tileSizeX; x++) {
                const int idxPix = x + (xa << 3) + y * (tileSizeX << (xCount - 1));
                if (mask[idxPix] == 255) continue;
                const size_t pi = (size_t)(px + x + (xa << 3)) + (size_t)(py + y) * w;
                const int rgb[3] = { e->plane[0][pi], e->plane[1][pi], e->plane[2][pi] };
                int m[3];
                for (int c = 0; c < 3; c++) {
                    float rel = (float)(rgb[c] - bb[c]);
                    if (d[c]) rel /= d[c];
                    rel *= 63.0f;
                    m[c] = ((mode >> c) & 1) ? (int)(63 - rel) : (int)rel;
                }
                swap3(mode >> 3, &m[0], &m[1], &m[2]);
                const int cell = m[0] + m[1] * 64 + (m[2] << 12);
                int lDiff[4];
                for (int k = 0; k < 4; k++) {                                  /* k = 0: 6 bit .. 3: 3 bit */
                    const int idx = P->pos[k][cell];
                    int co[3] = { P->fac[k][0][idx], P->fac[k][1][idx], P->fac[k][2][idx] };
                    for (int c = 0; c < 3; c++) if ((mode >> c) & 1) co[c] = LUT_FACTOR - co[c];
                    swap3(mode >> 3, &co[0], &co[1], &co[2]);
                    int worst = 0;
                    for (int c = 0; c < 3; c++) {
                        const int v = bb[c] + (co[c] * d[c]) / LUT_FACTOR;
                        tile[k][idxPix * 3 + c] = v;
                        const int df = abs(v - rgb[c]);
                        if (df > worst) worst = df;
                    }
                    lDiff[k] = worst;
                    absErr[k] += worst;
                    if (worst > 5) { wrong[k]++; if (wrong[k] > 3) reject |= 1 << k; }
                    P->value[k][streamIdx] = (uint8_t)idx;
                }
                (void)lDiff;
                streamIdx++;
            }
            if (reject == 0xF) return SKIP;                                     /* early exit: minDiff stays untouched (:6076-6079) */
        }
    }
    if ((reject & 1) == 0) { *minDiff = absErr[0]; res = M6; }
    if ((reject & 2) == 0) { *minDiff = absErr[1]; res = M5; }
    if ((reject & 4) == 0) { *minDiff = absErr[2]; res = M4; }
    if ((reject & 8) == 0) { *minDiff = absErr[3]; res = M3; }
    return res;
}

/* Correlation3DSearch (:6245-6781) for one tile shape; returns the number of matched tiles */
int yko_lut_search(yko_enc* e, int sx, int sy) {
    struct yko_lut_state* S = e->lut;
    if (!S || !S->tileType) return -1;
    const int w = e->w, h = e->h, TX = 1 << sx, TY = 1 << sy;
    int bigX, bigY, bitCount; swizzle_size_lut(sx, sy, &bigX, &bigY, &bitCount);
    int mapId = -1;
    if (TX == 16 && TY == 8) mapId = 0; else if (TX == 8 && TY == 16) mapId = 1; else if (TX == 8 && TY == 8) mapId = 2;
    else if (TX == 8 && TY == 4) mapId = 3; else if (TX == 4 && TY == 8) mapId = 4; else if (TX == 4 && TY == 4) mapId = 5;
    const int xBB = (w + bigX - 1) / bigX, stepY = bigX / TX;
    int matched = 0;
    static int tile[4][128 * 3];
    for (int sy0 = 0, posYS = 0; sy0 < h; sy0 += bigY, posYS += bitCount * xBB) {
        for (int sx0 = 0, posXS = posYS; sx0 < w; sx0 += bigX, posXS += bitCount) {
            int posY = posXS;
            for (int y = sy0; y < sy0 + bigY; y += TY, posY += stepY) {
                if (y >= h || y + TY > h) break;
                int pos = posY;
                for (int x = sx0; x < sx0 + bigX; x += TX, pos++) {
                    if (x >= w || x + TX > w) break;
                    /* buildBBox3D (:132-193): mask = pixels already covered in all three planes; box over the others */
                    uint8_t mask[128];
                    int bb[6] = { -1, -1, -1, -1, -1, -1 }, pixels = 0, first = 1;
                    for (int ty = 0; ty < TY; ty++) for (int tx = 0; tx < TX; tx++) {
                        const size_t pi = (size_t)(x + tx) + (size_t)(y + ty) * w;
                        if (e->mapSmoothTile[0][pi] == 255 && e->mapSmoothTile[1][pi] == 255 && e->mapSmoothTile[2][pi] == 255) { mask[tx + ty * TX] = 255; continue; }
                        mask[tx + ty * TX] = (uint8_t)pixels++;
                        for (int c = 0; c < 3; c++) {
                            const int v = e->plane[c][pi];
                            if (first || v < bb[c]) bb[c] = v;
                            if (first || v > bb[3 + c]) bb[3 + c] = v;
                        }
                        first = 0;
                    }
                    const int d[3] = { bb[3] - bb[0], bb[4] - bb[1], bb[5] - bb[2] };
                    const int accept = ((d[0] == 0) && d[1] != 0 && d[2] != 0) | ((d[1] == 0) && d[0] != 0 && d[2] != 0) | ((d[2] == 0) && d[0] != 0 && d[1] != 0) |
                                       (d[0] != 0 && d[1] != 0 && d[2] != 0);
                    if (!accept || pixels == 0) continue;
                    const int n[3] = { d[0] ? (1 << 20) / d[0] : 0, d[1] ? (1 << 20) / d[1] : 0, d[2] ? (1 << 20) / d[2] : 0 };
                    const float div = (float)(1 << 20);
                    for (int k = 0; k < S->nPat; k++) { memset(S->pat[k].sum, 0, sizeof S->pat[k].sum); S->pat[k].samples = 0; }
                    for (int ty = 0; ty < TY; ty++) for (int tx = 0; tx < TX; tx++) {
                        if (mask[tx + ty * TX] == 255) continue;
                        const size_t pi = (size_t)(x + tx) + (size_t)(y + ty) * w;
                        int i64[3];
                        for (int c = 0; c < 3; c++) { const int v = (e->plane[c][pi] - bb[c]) * n[c]; const float f = v / div; i64[c] = (int)(f * 63); }
                        for (int k = 0; k < S->nPat; k++) {
                            /* EvaluatePoint3D (EncoderContext.h:629-686): the axis swap of group n >> 3 is applied to the running x,y,z on
                             * EVERY iteration, so the 48 entries are cumulative permutations, not the six of swap3D */
                            lut_pattern* P = &S->pat[k];
                            int px = i64[0], py = i64[1], pz = i64[2];
                            for (int m = 0; m < 48; m++) {
                                swap3(m >> 3, &px, &py, &pz);
                                const int fx = (m & 1) ? 63 - px : px, fy = (m & 2) ? 63 - py : py, fz = (m & 4) ? 63 - pz : pz;
                                P->sum[m] += P->dist[fx + (fy << 6) + (fz << 12)];
                            }
                            P->samples++;
                        }
                    }
                    int found = 0, foundE = -1, foundM = -1, diffSum = (int)99999999999LL, bitMode = SKIP;
                    lut_pattern* best = NULL;
                    for (int k = 0; k < S->nPat; k++) {
                        lut_pattern* P = &S->pat[k];
                        int mode48 = -1; float minScore = 999999999.0f;                 /* GetEvaluation3D (:697-711) */
                        for (int f = 0; f < 48; f++) { const float avg = P->sum[f] / (float)(P->samples * 1024.0f); if (avg < minScore) { minScore = avg; mode48 = f; } }
                        int diffL = 0;
                        const int m = compute_values(e, P, TX, TY, mask, mode48, x, y, bb, &diffL, tile);
                        if (m != SKIP && diffL <= diffSum) { bitMode = m; found = 1; foundE = k; foundM = mode48; diffSum = diffL; best = P; }
                    }
                    if (!found) continue;
                    matched++;
                    for (int c = 0; c < 6; c++) S->color[S->nColor++] = (uint8_t)bb[c];       /* roundNBit(bb, 0) is the identity */
                    if (mapId >= 0) S->map[mapId][pos >> 3] |= (uint8_t)(1 << (pos & 7));
                    S->tileType[S->nType++] = (uint16_t)(foundM | (bitMode << 14) | (foundE << 6));
                    const int sel = 3 - bitMode;                                               /* value / factor tables: [0] = 6 bit */
                    memcpy(&S->idx[bitMode][S->nIdx[bitMode]], best->value[sel], (size_t)pixels);
                    S->nIdx[bitMode] += pixels;
                    /* the encoder's own rendering of the tile with the decoder's table (:6655-6737): entry table swapped per group and
                     * flipped per bit, colour = lo + (range * entry) / FACTOR */
                    {
                        const int maxIdx = (8 << bitMode) - 1;
                        int lut[3][64];
                        for (int i = 0; i <= maxIdx; i++) for (int c = 0; c < 3; c++) lut[c][i] = best->fac[sel][c][i];
                        int* lp[3] = { lut[0], lut[1], lut[2] }; int* t;
                        switch (foundM >> 3) {                                                  /* swap3DTable :5356-5390 */
                        case 1: t = lp[2]; lp[2] = lp[1]; lp[1] = t; break;
                        case 2: t = lp[0]; lp[0] = lp[1]; lp[1] = t; break;
                        case 3: t = lp[0]; lp[0] = lp[1]; lp[1] = lp[2]; lp[2] = t; break;
                        case 4: t = lp[1]; lp[1] = lp[0]; lp[0] = lp[2]; lp[2] = t; break;
                        case 5: t = lp[0]; lp[0] = lp[2]; lp[2] = t; break;
                        default: break;
                        }
                        const uint8_t* src = best->value[sel];
                        const int lX = TX > 8 ? 8 : TX, cX = TX > 8 ? 2 : 1;
                        for (int vx = 0; vx < cX; vx++) for (int ty = 0; ty < TY; ty++) for (int tx = 0; tx < lX; tx++) {
                            if (mask[tx + vx * 8 + ty * TX] == 255) continue;
                            const int i = *src++;
                            for (int c = 0; c < 3; c++) {
                                const int en = ((foundM >> c) & 1) ? LUT_FACTOR - lp[c][i] : lp[c][i];
                                S->preview[c][(size_t)(x + tx + vx * 8) + (size_t)(y + ty) * w] = bb[c] + (d[c] * en) / LUT_FACTOR;
                            }
                        }
                    }
                    for (int ty = 0; ty < TY; ty++) for (int tx = 0; tx < TX; tx++) {          /* the tile leaves the pool (:6760-6766) */
                        const size_t pi = (size_t)(x + tx) + (size_t)(y + ty) * w;
                        for (int c = 0; c < 3; c++) e->mapSmoothTile[c][pi] = 255;
                    }
                }
            }
        }
    }
    return matched;
}

const uint16_t* yko_lut_tile_types(const yko_enc* e, int* n) { *n = e->lut->nType; return e->lut->tileType; }
const uint8_t* yko_lut_colors(const yko_enc* e, int* n) { *n = e->lut->nColor; return e->lut->color; }
const uint8_t* yko_lut_indices(const yko_enc* e, int bits, int* n) { *n = e->lut->nIdx[bits - 3]; return e->lut->idx[bits - 3]; }
const uint8_t* yko_lut_map(const yko_enc* e, int which, int* n) { *n = e->lut->mapBytes[which]; return e->lut->map[which]; }
const int32_t* yko_lut_preview(const yko_enc* e, int plane) { return e->lut->preview[plane]; }

/* ------------------------------------------------------------------------------------------------------------------------------
 * Decoder side.  YAIK_AssignLUT (decoder/YAIK_API.cpp:133-415): the 'LUL0' file (LUTHeader, then per depth 3..6 and pattern the x, y, z
 * entry lists BinarySave3D wrote, EncoderContext.cpp:5452) becomes, per depth, [pattern][64 slots, 48 used = 6 axis orders x 8 flips]
 * [entry][3].  Tile3D_16x8 .. Tile3D_4x4 (decoder/YAIK_3DTile.cpp:244-2140): for every set bit of a pass's tile map, in scan order, pop
 * 6 colour bytes and a tile word (orientation | pattern << 6, depth in bits 14-15), then one pre-multiplied index per pixel of every 4x4
 * cell of the tile that tile4x4Mask does not mark yet -- 8x8 block by block (left before right, top before bottom), rows of a block
 * top down -- colour = lo + ((hi - lo) * entry >> 7); afterwards every cell of the tile is marked. */
int yko_lut_file(const yko_enc* e, uint8_t* out, int cap) {                  /* what RegisterAndCreate3DLut writes to LutFile.lut (:7820-7847) */
    const struct yko_lut_state* S = e->lut;
    const int n = 8 + (64 + 32 + 16 + 8) * 3 * S->nPat;
    if (!out) return n;
    if (cap < n) return -1;
    memset(out, 0, 8);
    out[0] = 'L'; out[1] = 'U'; out[2] = 'L'; out[3] = '0'; out[4] = 0; out[5] = (uint8_t)(S->nPat - 1); out[6] = 1;     /* padding_extension[0] = 0 then = 1 (:7824-7825) */
    uint8_t* w = out + 8;
    for (int depth = 3; depth >= 0; depth--)                                  /* file order: 3, 4, 5, 6 bit; fac[] order: 6, 5, 4, 3 */
        for (int k = 0; k < S->nPat; k++)
            for (int c = 0; c < 3; c++) for (int m = 0; m < (64 >> depth); m++) *w++ = (uint8_t)S->pat[k].fac[depth][c][m];
    return n;
}

int yko_dec_lut3d(yko_dec* d, const uint8_t* lutFile, int lutBytes, const uint8_t* const maps[6], const uint16_t* tiles, int nTiles,
                  const uint8_t* colors /* after PaletteFullRangeRemapping */, const uint8_t* const idx[4] /* 3,4,5,6 bit, x 3 */, int consumed[6]) {
    if (lutBytes < 8 || lutFile[0] != 'L' || lutFile[1] != 'U' || lutFile[2] != 'L') return -1;
    const int nPat = lutFile[5] + 1;
    if (lutBytes != 8 + nPat * 3 * (64 + 32 + 16 + 8)) return -1;
    const uint8_t* depthBase[4]; { const uint8_t* p = lutFile + 8; for (int b = 0; b < 4; b++) { depthBase[b] = p; p += (size_t)(8 << b) * 3 * nPat; } }
    static const int axis[6][3] = { {0,1,2}, {0,2,1}, {1,0,2}, {1,2,0}, {2,0,1}, {2,1,0} };
    static const int sz[6][2] = { {4,3}, {3,4}, {3,3}, {3,2}, {2,3}, {2,2} };
    int used[6] = { 0, 0, 0, 0, 0, 0 };                                       /* tiles, colour bytes, 3/4/5/6-bit indices */
    for (int k = 0; k < 6; k++) {
        const int sx = sz[k][0], sy = sz[k][1], TX = 1 << sx, TY = 1 << sy;
        int bigX, bigY, bitCount; swizzle_size_lut(sx, sy, &bigX, &bigY, &bitCount);
        const int xBB = (d->w + bigX - 1) / bigX, yBB = (d->h + bigY - 1) / bigY, tpr = bigX / TX;
        for (int by = 0; by < yBB; by++) for (int bx = 0; bx < xBB; bx++) for (int t = 0; t < bitCount; t++) {
            const int pos = (by * xBB + bx) * bitCount + t;
            if (!((maps[k][pos >> 3] >> (pos & 7)) & 1)) continue;
            const int x0 = bx * bigX + (t % tpr) * TX, y0 = by * bigY + (t / tpr) * TY;
            if (x0 >= d->w || y0 >= d->h || used[0] >= nTiles) continue;
            const uint8_t* RGB = colors + used[1]; used[1] += 6;
            const int tile = tiles[used[0]++], fmt = (tile >> 14) & 3, len = 8 << fmt;
            const int pattern = (tile >> 6) & 255, orient = tile & 63;
            const uint8_t* src = idx[fmt] + used[2 + fmt];
            int n = 0;
            const int diff[3] = { RGB[3] - RGB[0], RGB[4] - RGB[1], RGB[5] - RGB[2] };
            const int xCount = TX > 8 ? 2 : 1, lX = TX > 8 ? 8 : TX;
            for (int xa = 0; xa < xCount; xa++) for (int y = 0; y < TY; y++) for (int x = 0; x < lX; x++) {
                const int gx = x0 + x + xa * 8, gy = y0 + y, cx = gx >> 2, cy = gy >> 2;
                const int byteI = (cx >> 2) + (cy >> 1) * d->stride4, bit = (((cx >> 1) & 1) << 2) | ((cy & 1) << 1) | (cx & 1);
                if ((d->tile4x4Mask[byteI] >> bit) & 1) continue;
                const int e3 = src[n++];                                       /* entry number x 3 */
                const int entry = e3 / 3;
                for (int c = 0; c < 3; c++) {
                    int v = 251;                                                /* slots 48..63 and patterns beyond the file hold filler (:400-404) */
                    if (pattern < nPat && orient < 48 && entry < len) {
                        v = depthBase[fmt][(size_t)pattern * len * 3 + (size_t)axis[orient >> 3][c] * len + entry];
                        if ((orient >> c) & 1) v = 128 - v;
                    }
                    d->planes[(size_t)c * d->planeSize + ((size_t)(gy >> 3) * d->tileW + (gx >> 3)) * 64 + (gy & 7) * 8 + (gx & 7)] = (uint8_t)(RGB[c] + ((diff[c] * v) >> 7));
                }
            }
            used[2 + fmt] += n;
            for (int cy = y0 >> 2; cy < (y0 + TY) >> 2; cy++) for (int cx = x0 >> 2; cx < (x0 + TX) >> 2; cx++)
                d->tile4x4Mask[(cx >> 2) + (cy >> 1) * d->stride4] |= (uint8_t)(1 << ((((cx >> 1) & 1) << 2) | ((cy & 1) << 1) | (cx & 1)));
        }
    }
    used[0] *= 2;                                                              /* bytes of the tile stream, like the other counters */
    if (consumed) memcpy(consumed, used, sizeof used);
    return 0;
}
