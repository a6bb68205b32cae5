// yk_lut3d.hip — SURVEY 8(f)4: the 3-D LUT tile search of the encoder.
//   EncoderContext::Load3DPattern            encoder/EncoderContext.cpp:7851-7917   (sortPalette :2920-2960)
//   EvalCtx3D::Set3DPointCloud               :4744-4814
//   EncoderContext::Correlation3DSearch      :6245-6781   (buildBBox3D :132-193; EvalCtx3D::EvaluatePoint3D / GetEvaluation3D,
//                                                          encoder/EncoderContext.h:629-711; swap3D :5314-5354)
//   EncoderContext::computeValues3D          :5807-6094
// A tile of a pass (16x8, 8x16, 8x8, 8x4, 4x8, 4x4 in Convert()'s order, :9144-9199) whose still uncoded pixels, normalised into their own
// RGB bounding box, lie close to one of <= 64 point-cloud patterns under one of 48 axis permutations / flips is coded as (box, pattern,
// orientation, per-pixel entry index at 3..6 bits).  Tiles of one pass are disjoint and only read their own coverage, so a pass is
// embarrassingly parallel: ONE WORKGROUP PER TILE, then the same count / scan / emit compaction as the corner streams to put the
// results of the accepted tiles into the reference's streams in scan order.
//
// HBM layout of a pattern (built on the device by yk_lut_build_kernel): distance field u16[64^3] (the reference keeps int32; the largest
// squared distance is 3 * 63^2), nearest-entry tables u8[4][64^3] for 6 / 5 / 4 / 3 bits, factor tables s16[4][3][64], and the distance
// field ONCE PER ORIENTATION, orientation-minor: orient[cell][48] = dist[orientation_m(cell)], 24 MB per pattern.  The scoring reads, for one
// pixel, the 48 orientation distances of a pattern as 96 contiguous bytes (48 lanes, two cache lines) instead of 48 scattered 2-byte
// gathers: with 288 GB of HBM the 1.5 GB a full 64-pattern bank takes is the cheap side of that trade (DESIGN 3.8 has the measurements).
// Not on the timed path of bench.py.
#include "yk_common.h"
#include "yk_device.h"
#include <vector>

#define LUT_CUBE (64 * 64 * 64)
#define LUT_FACTOR 128                      // FACTOR, EncoderContext.cpp:22
#define LUT_MAXPAT 64

struct YkLutPattern { uint16_t* dist; uint16_t* orient; uint8_t* pos; int16_t* fac; int count; };      // device pointers
struct YkLutBank { const uint16_t* orient[LUT_MAXPAT]; const uint8_t* pos[LUT_MAXPAT]; const int16_t* fac[LUT_MAXPAT]; int nPat; };
struct YkLutState {
    YkLutPattern pat[LUT_MAXPAT]; int nPat = 0;
    YkLutBank* bankDev = nullptr;           // the table of pointers above, in HBM (too large for kernel arguments)
    bool started = false;
    // corr3D_* streams (StartCorrelationSearch :7316-7364), device
    uint16_t* tileType = nullptr; uint8_t* color = nullptr; uint8_t* idx[4] = {}; uint8_t* map[6] = {};
    size_t nType = 0, nColor = 0, nIdx[4] = {}, mapBytes[6] = {};
    size_t capTiles = 0, capPix = 0;
};

// the 48 orientations of EvaluatePoint3D: its axis swap of group n >> 3 is applied to the RUNNING x, y, z on every iteration of its loop
// over n (encoder/EncoderContext.h:629-686), so entry n is a cumulative permutation; table[n] = source axis of x | y << 2 | z << 4
__constant__ uint8_t c_lutPerm[48];
static void yk_lut_perm_table(uint8_t out[48]) {
    int a[3] = { 0, 1, 2 };
    for (int n = 0; n < 48; n++) {
        int t;
        switch (n >> 3) {                                                   // swap3D's cases on the running triple
        case 1: t = a[2]; a[2] = a[1]; a[1] = t; break;
        case 2: t = a[0]; a[0] = a[1]; a[1] = t; break;
        case 3: t = a[0]; a[0] = a[1]; a[1] = a[2]; a[2] = t; break;
        case 4: t = a[1]; a[1] = a[0]; a[0] = a[2]; a[2] = t; break;
        case 5: t = a[0]; a[0] = a[2]; a[2] = t; break;
        default: break;
        }
        out[n] = (uint8_t)(a[0] | (a[1] << 2) | (a[2] << 4));
    }
}
__device__ __forceinline__ void yk_lut_swap(int mode, int& x, int& y, int& z) {            // swap3D (:5314-5354), not cumulative
    int t;
    switch (mode) {
    case 1: t = z; z = y; y = t; break;
    case 2: t = x; x = y; y = t; break;
    case 3: t = x; x = y; y = z; z = t; break;
    case 4: t = y; y = x; x = z; z = t; break;
    case 5: t = x; x = z; z = t; break;
    default: break;
    }
}

// Set3DPointCloud's scan of the cube (:4786-4813): nearest point of every cell among every (1 << step)-th point, first minimum wins; the
// distance field is overwritten by every step, i.e. it ends as the distance to the nearest point of the 3-bit subset
__global__ __launch_bounds__(256) void yk_lut_build_kernel(const uint8_t* __restrict__ pts, int count, uint16_t* __restrict__ dist, uint8_t* __restrict__ pos) {
    __shared__ uint8_t s_pts[64 * 3];
    if (threadIdx.x < count * 3) s_pts[threadIdx.x] = pts[threadIdx.x];
    __syncthreads();
    const int i3 = blockIdx.x * blockDim.x + threadIdx.x;
    if (i3 >= LUT_CUBE) return;
    const int x = i3 & 63, y = (i3 >> 6) & 63, z = i3 >> 12;
#pragma unroll
    for (int step = 0; step < 4; step++) {
        int minDist = 999999999, best = 0;
        for (int p = 0; p < count; p += 1 << step) {
            const int dx = x - s_pts[p * 3], dy = y - s_pts[p * 3 + 1], dz = z - s_pts[p * 3 + 2];
            const int d = dx * dx + dy * dy + dz * dz;
            if (d < minDist) { minDist = d; best = p >> step; }
        }
        pos[(size_t)step * LUT_CUBE + i3] = (uint8_t)best;
        if (step == 3) dist[i3] = (uint16_t)minDist;
    }
}

// orient[cell * 48 + m] = dist at the cell orientation m maps `cell` to (the index EvaluatePoint3D forms for entry m)
__global__ __launch_bounds__(256) void yk_lut_orient_kernel(const uint16_t* __restrict__ dist, uint16_t* __restrict__ orient) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)LUT_CUBE * 48) return;
    const int cell = (int)(i / 48), m = (int)(i - (size_t)cell * 48);
    const int q[3] = { cell & 63, (cell >> 6) & 63, cell >> 12 };
    const int perm = c_lutPerm[m], ax = perm & 3, ay = (perm >> 2) & 3, az = (perm >> 4) & 3;
    const int fx = (m & 1) ? 63 - q[ax] : q[ax], fy = (m & 2) ? 63 - q[ay] : q[ay], fz = (m & 4) ? 63 - q[az] : q[az];
    orient[i] = dist[fx + (fy << 6) + (fz << 12)];
}

struct LutGeo { int sx, sy, bigX, bigY, bitCount, xBB, tilesPerRow, mapId; };
static LutGeo yk_lut_geo(int sx, int sy, int w) {
    LutGeo g; g.sx = sx; g.sy = sy;
    g.bigX = sx == 2 ? 32 : 64; g.bigY = sy == 2 ? 32 : 64;              // getSwizzleSize, include/YAIK_private.h:212-276
    g.tilesPerRow = g.bigX >> sx; g.bitCount = g.tilesPerRow * (g.bigY >> sy);
    g.xBB = (w + g.bigX - 1) / g.bigX;
    const int TX = 1 << sx, TY = 1 << sy;
    g.mapId = (TX == 16 && TY == 8) ? 0 : (TX == 8 && TY == 16) ? 1 : (TX == 8 && TY == 8) ? 2 : (TX == 8 && TY == 4) ? 3 : (TX == 4 && TY == 8) ? 4 : (TX == 4 && TY == 4) ? 5 : -1;
    return g;
}

// per tile slot of a pass: what the compaction needs
struct LutSlot { uint16_t type; uint8_t box[6]; uint8_t mode; uint8_t pixels; uint8_t found; uint8_t pad; };      // 12 bytes

// ---- one workgroup (128 threads) per tile ---------------------------------------------------------------------------------------
// thread t = the t-th pixel of the tile in the order computeValues3D walks it (left 8 columns first for 16-wide tiles, :5856-5859)
__global__ __launch_bounds__(128) void yk_lut_search_kernel(const int32_t* __restrict__ pR, const int32_t* __restrict__ pG, const int32_t* __restrict__ pB, int strideElems,
                                                            int w, int h, LutGeo g, const YkLutBank* __restrict__ bank, uint32_t* __restrict__ covCh32, size_t covStride,
                                                            int mtW, LutSlot* __restrict__ slots, uint8_t* __restrict__ slotIdx, uint32_t* __restrict__ bitmap) {
    __shared__ int s_i64[128];                                              // normalised 6-bit coordinates x | y << 6 | z << 12, -1 = masked pixel
    __shared__ int s_box[6], s_n;
    // sized by the bank at launch (a 64-pattern bank needs 19 KB, the usual handful 2 KB: four more workgroups per CU)
    extern __shared__ int s_dyn[];
    const int nPatS = bank->nPat;
    int* const s_sum = s_dyn;                                               // [nPat][48]
    int* const s_mode = s_sum + nPatS * 48;                                 // [nPat]
    int (*const s_acc)[8] = reinterpret_cast<int (*)[8]>(s_mode + nPatS);   // [nPat][8]: absErr of 6,5,4,3 bit; pixels with error > 5 of 6,5,4,3 bit
    int (*const s_part0)[8] = s_acc + nPatS;                                // the same per wave
    int (*const s_part1)[8] = s_part0 + nPatS;
    __shared__ int s_best[4];                                               // pattern, orientation, bit mode, found
    const int t = threadIdx.x, NT = blockDim.x, TX = 1 << g.sx, TY = 1 << g.sy, nPix = TX * TY;      // NT = 128, or 64 for tiles of at most 64 pixels
    const uint32_t pos = blockIdx.x;
    const uint32_t blk = pos / (uint32_t)g.bitCount, tt = pos % (uint32_t)g.bitCount;
    const int x0 = (int)(blk % (uint32_t)g.xBB) * g.bigX + (int)(tt % (uint32_t)g.tilesPerRow) * TX;
    const int y0 = (int)(blk / (uint32_t)g.xBB) * g.bigY + (int)(tt / (uint32_t)g.tilesPerRow) * TY;
    if (x0 + TX > w || y0 + TY > h) { if (t == 0) slots[pos].found = 0; return; }             // partial tiles are never tried (:6304, :6311)
    const int nPat = bank->nPat;
    if (t < 6) s_box[t] = t < 3 ? 9999 : -1;
    if (t == 0) s_n = 0;
    __syncthreads();
    // buildBBox3D (:132-193): a pixel is out when all three planes already cover it; the box spans the others
    int px = 0, py = 0; bool live = false; int v[3] = { 0, 0, 0 };
    if (t < nPix) {
        if (TX == 16) { px = (t & 7) + ((t / (8 * TY)) << 3); py = (t % (8 * TY)) >> 3; } else { px = t % TX; py = t / TX; }
        const int gx = x0 + px, gy = y0 + py;
        const size_t mt = (size_t)(gy >> 4) * mtW + (gx >> 4);
        const int cbit = ((gy >> 2) & 3) * 4 + ((gx >> 2) & 3);
        const uint16_t* cov = reinterpret_cast<const uint16_t*>(covCh32);
        live = !(((cov[mt] & cov[covStride + mt] & cov[2 * covStride + mt]) >> cbit) & 1);
        if (live) {
            const size_t pi = (size_t)gy * strideElems + gx;
            v[0] = pR[pi]; v[1] = pG[pi]; v[2] = pB[pi];
#pragma unroll
            for (int c = 0; c < 3; c++) { atomicMin(&s_box[c], v[c]); atomicMax(&s_box[3 + c], v[c]); }
            atomicAdd(&s_n, 1);
        }
    }
    __syncthreads();
    const int pixels = s_n;
    int lo[3], d[3];
#pragma unroll
    for (int c = 0; c < 3; c++) { lo[c] = s_box[c]; d[c] = s_box[3 + c] - s_box[c]; }
    const bool accept = pixels != 0 && (((d[0] == 0) && d[1] != 0 && d[2] != 0) || ((d[1] == 0) && d[0] != 0 && d[2] != 0) || ((d[2] == 0) && d[0] != 0 && d[1] != 0) ||
                                        (d[0] != 0 && d[1] != 0 && d[2] != 0));                 // :6322-6326
    if (!accept || nPat == 0) { if (t == 0) slots[pos].found = 0; return; }
    {   // coordinates for the scoring (:6389-6405): (int)(((v - lo) * ((1 << 20) / d)) / 2^20 * 63) in float
        int cell = -1;
        if (live) {
            int q[3];
#pragma unroll
            for (int c = 0; c < 3; c++) { const int n = d[c] ? (1 << 20) / d[c] : 0; const float f = __fdiv_rn((float)((v[c] - lo[c]) * n), 1048576.0f); q[c] = (int)__fmul_rn(f, 63.0f); }
            cell = q[0] | (q[1] << 6) | (q[2] << 12);
        }
        s_i64[t] = cell;
    }
    __syncthreads();
    // EvaluatePoint3D: sum of the distance field over the tile's pixels for every (pattern, orientation).  Threads 0..95 = two patterns x 48
    // orientations per round; for one pixel the 48 lanes of a pattern read 96 contiguous bytes of its orientation-minor table.
    const int perRound = NT >= 96 ? 2 : 1;
    for (int k0 = 0; k0 < nPat; k0 += perRound) {
        const int k = k0 + t / 48, m = t % 48;
        if (t < 48 * perRound && k < nPat) {
            const uint16_t* __restrict__ orient = bank->orient[k];
            // branch-free and eight pixels at a time: the loads of a thread are independent, a loop of dependent-latency round trips (one
            // load per iteration behind a test) was the whole cost of this kernel; s_i64 holds -1 beyond the tile's pixels
            int sum = 0;
            for (int p0 = 0; p0 < nPix; p0 += 8) {
                int cell[8]; uint32_t d8[8];
#pragma unroll
                for (int j = 0; j < 8; j++) cell[j] = s_i64[p0 + j];
#pragma unroll
                for (int j = 0; j < 8; j++) d8[j] = orient[(size_t)max(cell[j], 0) * 48 + m];
#pragma unroll
                for (int j = 0; j < 8; j++) sum += cell[j] >= 0 ? (int)d8[j] : 0;
            }
            s_sum[k * 48 + m] = sum;
        }
    }
    __syncthreads();
    // GetEvaluation3D (:697-711): first minimum of sum / (samples * 1024.0f) in float
    for (int k = t; k < nPat; k += NT) {
        int res = -1; float minScore = 999999999.0f;
        const float den = __fmul_rn((float)pixels, 1024.0f);
        for (int f = 0; f < 48; f++) { const float avg = __fdiv_rn((float)s_sum[k * 48 + f], den); if (avg < minScore) { minScore = avg; res = f; } }
        s_mode[k] = res;
    }
    __syncthreads();
    // computeValues3D for every pattern at its best orientation: per pixel the entry at 6 / 5 / 4 / 3 bits and its worst channel error.
    // Its early exit (all four depths rejected at the end of a row) returns what the full pass returns, so the sums are order-free.
    // the pixel's position in the box (:5871-5887), the same for every pattern and depth; lanes without a pixel sit on the box's low corner
    float relp[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float rel = live ? (float)(v[c] - lo[c]) : 0.0f;
        if (d[c]) rel = __fdiv_rn(rel, (float)d[c]);
        relp[c] = __fmul_rn(rel, 63.0f);
    }
    auto entry = [&](int k, int mode, int depth /*0 = 6 bit*/, int& idx, int (&col)[3]) {
        int m[3];
#pragma unroll
        for (int c = 0; c < 3; c++) m[c] = ((mode >> c) & 1) ? (int)__fsub_rn(63.0f, relp[c]) : (int)relp[c];
        yk_lut_swap(mode >> 3, m[0], m[1], m[2]);
        idx = bank->pos[k][(size_t)depth * LUT_CUBE + (m[0] + m[1] * 64 + (m[2] << 12))];
        const int16_t* __restrict__ fac = bank->fac[k] + depth * 3 * 64;
        int co[3] = { fac[idx], fac[64 + idx], fac[128 + idx] };
#pragma unroll
        for (int c = 0; c < 3; c++) if ((mode >> c) & 1) co[c] = LUT_FACTOR - co[c];
        yk_lut_swap(mode >> 3, co[0], co[1], co[2]);
#pragma unroll
        for (int c = 0; c < 3; c++) col[c] = lo[c] + (co[c] * d[c]) / LUT_FACTOR;
    };
    // per pattern: the four depths' worst-channel errors of this pixel, summed over the tile with wave reductions (two 16-bit sums per
    // shuffle word: a tile holds at most 128 pixels of error <= 255) and ballots for the ">5" counts; LDS atomics on eight shared words
    // per pattern serialised all 64 lanes of a wave and were the larger half of this kernel's time
    {
        const int wv = t >> 6;
        auto evalPattern = [&](const int k, int (&w4)[4]) {
            const int mode = s_mode[k];
#pragma unroll
            for (int depth = 0; depth < 4; depth++) {
                int idx, col[3];
                entry(k, mode, depth, idx, col);
                w4[depth] = live ? max(max(abs(col[0] - v[0]), abs(col[1] - v[1])), abs(col[2] - v[2])) : 0;
            }
        };
        auto reducePattern = [&](const int k, const int (&w4)[4]) {
            int s01 = w4[0] | (w4[1] << 16), s23 = w4[2] | (w4[3] << 16);
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) { s01 += __shfl_xor(s01, off); s23 += __shfl_xor(s23, off); }
            const int c0 = __popcll(__ballot(w4[0] > 5)), c1 = __popcll(__ballot(w4[1] > 5)), c2 = __popcll(__ballot(w4[2] > 5)), c3 = __popcll(__ballot(w4[3] > 5));
            if ((t & 63) == 0) {
                int* a = (wv ? s_part1 : s_part0)[k];
                a[0] = s01 & 0xFFFF; a[1] = s01 >> 16; a[2] = s23 & 0xFFFF; a[3] = s23 >> 16; a[4] = c0; a[5] = c1; a[6] = c2; a[7] = c3;
            }
        };
        // two patterns per round: their eight look-up chains (cell -> entry -> three factors) are in flight together.  Lanes without a pixel
        // run the same loads on a harmless cell (every lane of a wave takes part in the reductions) and contribute 0.
        for (int k = 0; k < nPat; k += 2) {
            int wa[4], wb[4] = { 0, 0, 0, 0 };
            evalPattern(k, wa);
            if (k + 1 < nPat) evalPattern(k + 1, wb);
            reducePattern(k, wa);
            if (k + 1 < nPat) reducePattern(k + 1, wb);
        }
    }
    __syncthreads();
    for (int i = t; i < nPat * 8; i += NT) s_acc[i >> 3][i & 7] = s_part0[i >> 3][i & 7] + (NT > 64 ? s_part1[i >> 3][i & 7] : 0);
    __syncthreads();
    if (t == 0) {
        // :6066-6069 (lowest depth that is not rejected, a depth is rejected when more than 3 pixels are off by more than 5) and the choice
        // among patterns :6486 (smallest summed error, the LATER pattern on a tie)
        int found = 0, bestK = -1, bestMode = 4, diffSum = (int)99999999999LL;
        for (int k = 0; k < nPat; k++) {
            int res = 4, diff = 0;
            if (s_acc[k][4] <= 3) { diff = s_acc[k][0]; res = 3; }
            if (s_acc[k][5] <= 3) { diff = s_acc[k][1]; res = 2; }
            if (s_acc[k][6] <= 3) { diff = s_acc[k][2]; res = 1; }
            if (s_acc[k][7] <= 3) { diff = s_acc[k][3]; res = 0; }
            if (res != 4 && diff <= diffSum) { found = 1; bestK = k; bestMode = res; diffSum = diff; }
        }
        s_best[0] = bestK; s_best[1] = found ? s_mode[bestK] : 0; s_best[2] = bestMode; s_best[3] = found;
        LutSlot sl;
        sl.found = (uint8_t)found; sl.pixels = (uint8_t)pixels; sl.mode = (uint8_t)bestMode; sl.pad = 0;
        sl.type = (uint16_t)((found ? s_mode[bestK] : 0) | (bestMode << 14) | ((found ? bestK : 0) << 6));        // :6559
        for (int c = 0; c < 6; c++) sl.box[c] = (uint8_t)s_box[c];
        slots[pos] = sl;
        if (found) atomicOr(&bitmap[pos >> 5], 1u << (pos & 31));
    }
    __syncthreads();
    if (!s_best[3]) return;
    // the winner's indices in stream order: rank of the pixel among the live ones (thread order = the reference's walk)
    const unsigned long long bal = __ballot(live);
    __shared__ int s_w0;
    if (t == 0) s_w0 = __popcll(bal);
    __syncthreads();
    if (live) {
        const int rank = __popcll(bal & ((1ULL << (t & 63)) - 1ULL)) + (t >= 64 ? s_w0 : 0);      // s_w0: live pixels of the first wave
        int idx, col[3];
        entry(s_best[0], s_best[1], 3 - s_best[2], idx, col);
        slotIdx[(size_t)pos * nPix + rank] = (uint8_t)idx;
    }
    // the tile leaves the pool: all three planes, every cell of the tile (:6760-6766); cells of other tiles share words -> atomics
    if (t < (TX >> 2) * (TY >> 2)) {
        const int cx = (x0 >> 2) + t % (TX >> 2), cy = (y0 >> 2) + t / (TX >> 2);
        const size_t mt = (size_t)(cy >> 2) * mtW + (cx >> 2);
        const int cbit = (cy & 3) * 4 + (cx & 3);
#pragma unroll
        for (int n = 0; n < 3; n++) { const size_t e = n * covStride + mt; atomicOr(&covCh32[e >> 1], 1u << (cbit + 16 * (e & 1))); }
    }
}

// ---- compaction in scan order: thread per tile slot, 1024 slots per workgroup; five running counts (tiles, index bytes of 3/4/5/6 bit) ----
__device__ __forceinline__ void yk_lut_counts(const LutSlot& s, bool valid, uint32_t (&cnt)[5]) {
    const bool f = valid && s.found;
    cnt[0] = f ? 1u : 0u;
#pragma unroll
    for (int m = 0; m < 4; m++) cnt[1 + m] = (f && s.mode == m) ? (uint32_t)s.pixels : 0u;
}
__global__ __launch_bounds__(1024) void yk_lut_count_kernel(const LutSlot* __restrict__ slots, size_t nSlots, uint32_t* __restrict__ blockSums, size_t nb) {
    __shared__ uint32_t s_tmp[32];
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    LutSlot s = {}; if (i < nSlots) s = slots[i];
    uint32_t cnt[5]; yk_lut_counts(s, i < nSlots, cnt);
    for (int j = 0; j < 5; j++) { uint32_t tot; yk_block_exscan(cnt[j], s_tmp, &tot); if (threadIdx.x == 0) blockSums[j * nb + blockIdx.x] = tot; }
}
__global__ __launch_bounds__(1024) void yk_lut_scan_kernel(uint32_t* __restrict__ blockSums, size_t nb, uint32_t* __restrict__ totals) {
    __shared__ uint32_t s_tmp[32];
    for (int j = 0; j < 5; j++) {
        uint32_t base = 0;
        for (size_t start = 0; start < nb; start += 1024) {
            const size_t i = start + threadIdx.x;
            const uint32_t v = i < nb ? blockSums[j * nb + i] : 0u;
            uint32_t tot;
            const uint32_t e = yk_block_exscan(v, s_tmp, &tot);
            if (i < nb) blockSums[j * nb + i] = base + e;
            base += tot;
        }
        if (threadIdx.x == 0) totals[j] = base;
    }
}
struct LutStreams { uint16_t* tileType; uint8_t* color; uint8_t* idx[4]; unsigned long long nType, nColor, nIdx[4]; };
__global__ __launch_bounds__(1024) void yk_lut_emit_kernel(const LutSlot* __restrict__ slots, const uint8_t* __restrict__ slotIdx, int nPix, size_t nSlots,
                                                           const uint32_t* __restrict__ blockSums, size_t nb, LutStreams out) {
    __shared__ uint32_t s_tmp[32];
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    LutSlot s = {}; if (i < nSlots) s = slots[i];
    uint32_t cnt[5], ex[5]; yk_lut_counts(s, i < nSlots, cnt);
    for (int j = 0; j < 5; j++) { uint32_t tot; ex[j] = blockSums[j * nb + blockIdx.x] + yk_block_exscan(cnt[j], s_tmp, &tot); }
    if (!cnt[0]) return;
    const size_t ti = out.nType + ex[0];
    out.tileType[ti] = s.type;
    for (int c = 0; c < 6; c++) out.color[out.nColor + (size_t)ex[0] * 6 + c] = s.box[c];
    uint8_t* dst = out.idx[s.mode] + out.nIdx[s.mode] + ex[1 + s.mode];
    const uint8_t* src = slotIdx + i * nPix;
    for (int p = 0; p < s.pixels; p++) dst[p] = src[p];
}

static void yk_lut_release(yk_ctx* c) {
    YkLutState* S = c->lut; if (!S) return;
    auto F = [](auto*& p) { if (p) { (void)hipFree((void*)p); p = nullptr; } };
    for (int k = 0; k < S->nPat; k++) { F(S->pat[k].dist); F(S->pat[k].orient); F(S->pat[k].pos); F(S->pat[k].fac); }
    F(S->bankDev); F(S->tileType); F(S->color);
    for (auto& p : S->idx) F(p);
    for (auto& p : S->map) F(p);
    delete S; c->lut = nullptr;
}
void yk_lut_destroy(yk_ctx* c) { yk_lut_release(c); }

static uint32_t yk_morton3(int r, int g, int b) {                              // morton256_x | _y | _z (:2799-2912): bit k of r, g, b -> bits 3k, 3k+1, 3k+2
    uint32_t m = 0;
    for (int k = 0; k < 8; k++) m |= (uint32_t)((r >> k) & 1) << (3 * k) | (uint32_t)((g >> k) & 1) << (3 * k + 1) | (uint32_t)((b >> k) & 1) << (3 * k + 2);
    return m;
}

extern "C" {

int yk_lut_clear(yk_ctx* c) { if (!c) return YK_ERR_BAD_ARG; YK_HIP(c, hipSetDevice(c->device)); YK_HIP(c, hipStreamSynchronize(c->stream)); yk_lut_release(c); return YK_OK; }

int yk_lut_load_pattern(yk_ctx* c, const uint8_t* r, const uint8_t* g, const uint8_t* b, int count, int* index) {
    if (!c || !r || !g || !b) return YK_ERR_BAD_ARG;
    // more than 64 points make the reference read and write past its 64-entry tables (:7907-7917): refused
    if (count < 1 || count > 64) return yk_fail(c, YK_ERR_BAD_ARG, "a pattern holds 1..64 points");
    for (int n = 0; n < count; n++) if (r[n] > 63 || g[n] > 63 || b[n] > 63) return yk_fail(c, YK_ERR_BAD_ARG, "pattern coordinates are 6 bits");
    YK_HIP(c, hipSetDevice(c->device));
    if (!c->lut) {
        c->lut = new YkLutState();
        uint8_t perm[48]; yk_lut_perm_table(perm);
        YK_HIP(c, hipMemcpyToSymbol(HIP_SYMBOL(c_lutPerm), perm, sizeof perm));
    }
    YkLutState* S = c->lut;
    if (S->nPat >= LUT_MAXPAT) return yk_fail(c, YK_ERR_RANGE, "LUT 3D more than 64 entries");     // :7912
    uint8_t pts[64 * 3];
    for (int n = 0; n < count; n++) { pts[n * 3] = r[n]; pts[n * 3 + 1] = g[n]; pts[n * 3 + 2] = b[n]; }
    for (int i = 0; i < count - 1; i++) {                                      // sortPalette: selection sort on the morton code, first minimum
        int mn = i;
        for (int j = i + 1; j < count; j++)
            if (yk_morton3(pts[mn * 3], pts[mn * 3 + 1], pts[mn * 3 + 2]) > yk_morton3(pts[j * 3], pts[j * 3 + 1], pts[j * 3 + 2])) mn = j;
        if (mn != i) for (int k = 0; k < 3; k++) std::swap(pts[i * 3 + k], pts[mn * 3 + k]);
    }
    int16_t fac[4][3][64] = {};                                                // Set3DPointCloud :4749-4782
    for (int step = 0; step < 4; step++)
        for (int p = 0; p < count; p += 1 << step)
            for (int k = 0; k < 3; k++) fac[step][k][p >> step] = (int16_t)((pts[p * 3 + k] / 63.0f) * LUT_FACTOR);
    YkLutPattern& P = S->pat[S->nPat];
    P.count = count;
    YK_HIP(c, hipMalloc(&P.dist, LUT_CUBE * sizeof(uint16_t)));
    YK_HIP(c, hipMalloc(&P.orient, (size_t)LUT_CUBE * 48 * sizeof(uint16_t)));
    YK_HIP(c, hipMalloc(&P.pos, 4 * LUT_CUBE));
    YK_HIP(c, hipMalloc(&P.fac, sizeof fac));
    uint8_t* dPts = nullptr;
    YK_HIP(c, hipMalloc(&dPts, 64 * 3));
    YK_HIP(c, hipMemcpyAsync(dPts, pts, (size_t)count * 3, hipMemcpyHostToDevice, c->stream));
    YK_HIP(c, hipMemcpyAsync(P.fac, fac, sizeof fac, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(yk_lut_build_kernel, dim3(LUT_CUBE / 256), dim3(256), 0, c->stream, dPts, count, P.dist, P.pos);
    hipLaunchKernelGGL(yk_lut_orient_kernel, dim3((unsigned)(((size_t)LUT_CUBE * 48 + 255) / 256)), dim3(256), 0, c->stream, P.dist, P.orient);
    YK_HIP(c, hipGetLastError());
    YK_HIP(c, hipStreamSynchronize(c->stream));
    (void)hipFree(dPts);
    if (index) *index = S->nPat;
    S->nPat++;
    YkLutBank bank = {};
    for (int k = 0; k < S->nPat; k++) { bank.orient[k] = S->pat[k].orient; bank.pos[k] = S->pat[k].pos; bank.fac[k] = S->pat[k].fac; }
    bank.nPat = S->nPat;
    if (!S->bankDev) YK_HIP(c, hipMalloc(&S->bankDev, sizeof(YkLutBank)));
    YK_HIP(c, hipMemcpy(S->bankDev, &bank, sizeof bank, hipMemcpyHostToDevice));
    return YK_OK;
}

int yk_lut_pattern_tables(yk_ctx* c, int pattern, int16_t* factors /*4*3*64*/, uint16_t* distanceField /*64^3*/, uint8_t* positions /*4*64^3*/) {
    if (!c || !c->lut || pattern < 0 || pattern >= c->lut->nPat) return YK_ERR_BAD_ARG;
    YK_HIP(c, hipSetDevice(c->device));
    const YkLutPattern& P = c->lut->pat[pattern];
    if (factors) YK_HIP(c, hipMemcpy(factors, P.fac, 4 * 3 * 64 * sizeof(int16_t), hipMemcpyDeviceToHost));
    if (distanceField) YK_HIP(c, hipMemcpy(distanceField, P.dist, LUT_CUBE * sizeof(uint16_t), hipMemcpyDeviceToHost));
    if (positions) YK_HIP(c, hipMemcpy(positions, P.pos, 4 * LUT_CUBE, hipMemcpyDeviceToHost));
    return YK_OK;
}

int yk_lut_start(yk_ctx* c) {
    if (!c) return YK_ERR_BAD_ARG;
    if (!c->encoded) return yk_fail(c, YK_ERR_STATE, "yk_encode_tiles first (the 3-D LUT search follows the gradient passes)");
    if (!c->lut || c->lut->nPat == 0) return yk_fail(c, YK_ERR_STATE, "yk_lut_load_pattern first");
    if (c->nFrames != 1 || c->y0 != 0 || c->h != c->fullH) return yk_fail(c, YK_ERR_STATE, "the 3-D LUT search works on single whole images");
    YK_HIP(c, hipSetDevice(c->device));
    YkLutState* S = c->lut;
    const int w = c->fullW, h = c->h;
    auto F = [](auto*& p) { if (p) { (void)hipFree((void*)p); p = nullptr; } };
    F(S->tileType); F(S->color); for (auto& p : S->idx) F(p); for (auto& p : S->map) F(p);
    S->capTiles = (size_t)(w / 4) * (h / 4) + 16; S->capPix = (size_t)w * h + 128;
    YK_HIP(c, hipMalloc(&S->tileType, S->capTiles * 2));
    YK_HIP(c, hipMalloc(&S->color, S->capTiles * 6));
    for (auto& p : S->idx) YK_HIP(c, hipMalloc(&p, S->capPix));
    static const int sz[6][2] = { {4,3}, {3,4}, {3,3}, {3,2}, {2,3}, {2,2} };
    for (int k = 0; k < 6; k++) {
        const LutGeo g = yk_lut_geo(sz[k][0], sz[k][1], w);
        S->mapBytes[k] = (size_t)g.xBB * ((h + g.bigY - 1) / g.bigY) * g.bitCount;       // BitmapSwizzleMapSize (:7310): a bit count used as the byte size
        YK_HIP(c, hipMalloc(&S->map[k], S->mapBytes[k] + 16));
        YK_HIP(c, hipMemsetAsync(S->map[k], 0, S->mapBytes[k] + 16, c->stream));
    }
    S->nType = S->nColor = 0; for (auto& n : S->nIdx) n = 0;
    { int rc = yk_pp_activate(c); if (rc) return rc; }                          // LUT tiles paint mapSmoothTile only, never smoothMap
    S->started = true;
    return YK_OK;
}

int yk_lut_search(yk_ctx* c, int shiftX, int shiftY, int* matched) {
    if (!c) return YK_ERR_BAD_ARG;
    if (!c->lut || !c->lut->started) return yk_fail(c, YK_ERR_STATE, "yk_lut_start first");
    if (!c->encoded || !c->ppActive) return yk_fail(c, YK_ERR_STATE, "a new encode ended the search: yk_lut_start again");
    const LutGeo g = yk_lut_geo(shiftX, shiftY, c->fullW);
    if (g.mapId < 0) return yk_fail(c, YK_ERR_BAD_ARG, "tile shapes: 16x8, 8x16, 8x8, 8x4, 4x8, 4x4");
    YK_HIP(c, hipSetDevice(c->device));
    YkLutState* S = c->lut;
    const int w = c->fullW, h = c->h, nPix = (1 << shiftX) * (1 << shiftY);
    const size_t nSlots = (size_t)g.xBB * ((h + g.bigY - 1) / g.bigY) * g.bitCount, nb = (nSlots + 1023) / 1024;
    LutSlot* slots = nullptr; uint8_t* slotIdx = nullptr; uint32_t* sums = nullptr;
    YK_HIP(c, hipMalloc(&slots, nSlots * sizeof(LutSlot)));
    YK_HIP(c, hipMalloc(&slotIdx, nSlots * nPix));
    YK_HIP(c, hipMalloc(&sums, (5 * nb + 16) * sizeof(uint32_t)));
    { int rc = yk_stage_begin(c, YK_STAGE_LUT3D); if (rc) return rc; }
    hipLaunchKernelGGL(yk_lut_search_kernel, dim3((unsigned)nSlots), dim3(nPix > 64 ? 128 : 64), (size_t)S->nPat * (48 + 1 + 24) * sizeof(int), c->stream, c->plane[0], c->plane[1], c->plane[2], c->strideElems, w, h, g,
                       S->bankDev, reinterpret_cast<uint32_t*>(c->covCh), c->covChStride, c->mtW, slots, slotIdx, reinterpret_cast<uint32_t*>(S->map[g.mapId]));
    { int rc = yk_stage_end(c, YK_STAGE_LUT3D); if (rc) return rc; }
    hipLaunchKernelGGL(yk_lut_count_kernel, dim3((unsigned)nb), dim3(1024), 0, c->stream, slots, nSlots, sums, nb);
    hipLaunchKernelGGL(yk_lut_scan_kernel, dim3(1), dim3(1024), 0, c->stream, sums, nb, sums + 5 * nb);
    LutStreams out; out.tileType = S->tileType; out.color = S->color; out.nType = S->nType; out.nColor = S->nColor;
    for (int m = 0; m < 4; m++) { out.idx[m] = S->idx[m]; out.nIdx[m] = S->nIdx[m]; }
    hipLaunchKernelGGL(yk_lut_emit_kernel, dim3((unsigned)nb), dim3(1024), 0, c->stream, slots, slotIdx, nPix, nSlots, sums, nb, out);
    hipError_t e = hipGetLastError();
    uint32_t tot[5] = {};
    if (e == hipSuccess) e = hipMemcpyAsync(tot, sums + 5 * nb, sizeof tot, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(slots); (void)hipFree(slotIdx); (void)hipFree(sums);
    if (e != hipSuccess) return yk_fail(c, YK_ERR_HIP, "3-D LUT search", e);
    S->nType += tot[0]; S->nColor += (size_t)tot[0] * 6;
    for (int m = 0; m < 4; m++) S->nIdx[m] += tot[1 + m];
    c->r1Ready = false;
    if (matched) *matched = (int)tot[0];
    return YK_OK;
}

// which: 0 = tile types (u16), 1 = box colours (6 bytes per tile, before CompressF), 2..5 = entry indices of the 3 / 4 / 5 / 6 bit tiles (raw:
// the '3DTL' chunk stores them times 3, :7526), 6..11 = tile maps of 16x8, 8x16, 8x8, 8x4, 4x8, 4x4
int yk_lut_stream(yk_ctx* c, int which, uint8_t* hostOut, size_t cap, size_t* nBytes) {
    if (!c || which < 0 || which > 11) return YK_ERR_BAD_ARG;
    if (!c->lut || !c->lut->started) return yk_fail(c, YK_ERR_STATE, "yk_lut_start first");
    YkLutState* S = c->lut;
    const void* src; size_t n;
    if (which == 0) { src = S->tileType; n = S->nType * 2; }
    else if (which == 1) { src = S->color; n = S->nColor; }
    else if (which <= 5) { src = S->idx[which - 2]; n = S->nIdx[which - 2]; }
    else { src = S->map[which - 6]; n = S->mapBytes[which - 6]; }
    if (nBytes) *nBytes = n;
    if (hostOut && n) {
        if (cap < n) return yk_fail(c, YK_ERR_RANGE, "stream buffer too small");
        YK_HIP(c, hipSetDevice(c->device));
        YK_HIP(c, hipMemcpyAsync(hostOut, src, n, hipMemcpyDeviceToHost, c->stream));
        YK_HIP(c, hipStreamSynchronize(c->stream));
    }
    return YK_OK;
}

}  // extern "C"

// ==== decoder side =================================================================================================================
// YAIK_AssignLUT (decoder/YAIK_API.cpp:133-415) and Tile3D_16x8 .. Tile3D_4x4 (decoder/YAIK_3DTile.cpp:244-2140).  The reference walks the
// tile maps sequentially, popping 6 colour bytes, a tile word and one pre-multiplied index per still unmarked pixel per tile.  Here a pass
// is three small launches: rank of every set map bit (= position in the tile / colour streams), index bytes per depth of every tile from
// its tile word and the tile4x4Mask state (-> offsets in the four index streams), then the fill.  Tiles of a pass are disjoint.
struct YkLutDecState { uint8_t* tbl[4] = {}; int nPat = 0; };

__device__ __forceinline__ bool yk_dl_tile(const uint32_t* __restrict__ map, size_t nSlots, size_t pos, const LutGeo& g, int w, int h, int& x0, int& y0) {
    if (pos >= nSlots || !((map[pos >> 5] >> (pos & 31)) & 1u)) return false;
    const uint32_t blk = (uint32_t)pos / (uint32_t)g.bitCount, t = (uint32_t)pos % (uint32_t)g.bitCount;
    x0 = (int)(blk % (uint32_t)g.xBB) * g.bigX + (int)(t % (uint32_t)g.tilesPerRow) * (1 << g.sx);
    y0 = (int)(blk / (uint32_t)g.xBB) * g.bigY + (int)(t / (uint32_t)g.tilesPerRow) * (1 << g.sy);
    return x0 < w && y0 < h;
}
__device__ __forceinline__ bool yk_dl_marked(const uint8_t* __restrict__ tile4, int stride4, int gx, int gy) {
    const int cx = gx >> 2, cy = gy >> 2;
    return (tile4[(cx >> 2) + (cy >> 1) * stride4] >> ((((cx >> 1) & 1) << 2) | ((cy & 1) << 1) | (cx & 1))) & 1;
}
__global__ __launch_bounds__(1024) void yk_dl_rank_kernel(const uint32_t* __restrict__ map, size_t nSlots, LutGeo g, int w, int h, uint32_t* __restrict__ blockSums) {
    __shared__ uint32_t s_tmp[32];
    int x0, y0; uint32_t tot;
    yk_block_exscan(yk_dl_tile(map, nSlots, (size_t)blockIdx.x * 1024 + threadIdx.x, g, w, h, x0, y0) ? 1u : 0u, s_tmp, &tot);
    if (threadIdx.x == 0) blockSums[blockIdx.x] = tot;
}
// FILL = false: index bytes per depth and workgroup; FILL = true: decode
template <bool FILL>
__global__ __launch_bounds__(1024) void yk_dl_tile_kernel(const uint32_t* __restrict__ map, size_t nSlots, LutGeo g, int w, int h, const uint32_t* __restrict__ rankBase, size_t nb,
                                                          uint32_t* __restrict__ byteSums /*[4][nb]*/, const uint16_t* __restrict__ tiles, unsigned long long tileBase, unsigned long long nTiles,
                                                          const uint8_t* __restrict__ colors, const uint8_t* const __restrict__ idx0, const uint8_t* const __restrict__ idx1,
                                                          const uint8_t* const __restrict__ idx2, const uint8_t* const __restrict__ idx3, unsigned long long ib0, unsigned long long ib1,
                                                          unsigned long long ib2, unsigned long long ib3, const uint8_t* const __restrict__ t0, const uint8_t* const __restrict__ t1,
                                                          const uint8_t* const __restrict__ t2, const uint8_t* const __restrict__ t3, int nPat,
                                                          uint8_t* __restrict__ planes, size_t planeSize, int tileW, uint8_t* __restrict__ tile4, int stride4) {
    __shared__ uint32_t s_tmp[32];
    const size_t pos = (size_t)blockIdx.x * 1024 + threadIdx.x;
    int x0 = 0, y0 = 0;
    const bool set = yk_dl_tile(map, nSlots, pos, g, w, h, x0, y0);
    uint32_t tot;
    const uint32_t rank = rankBase[blockIdx.x] + yk_block_exscan(set ? 1u : 0u, s_tmp, &tot);
    const unsigned long long ti = tileBase + rank;
    const bool live = set && ti < nTiles;                                       // the reference trusts the counts of the header (its TODOs at :1079, :1108)
    const int TX = 1 << g.sx, TY = 1 << g.sy;
    int fmt = 0, tile = 0; uint32_t nbytes = 0;
    if (live) {
        tile = tiles[ti]; fmt = (tile >> 14) & 3;
        for (int cy = 0; cy < TY; cy += 4) for (int cx = 0; cx < TX; cx += 4) if (!yk_dl_marked(tile4, stride4, x0 + cx, y0 + cy)) nbytes += 16;
    }
    uint32_t off[4];
    for (int f = 0; f < 4; f++) {
        const uint32_t e = yk_block_exscan((live && fmt == f) ? nbytes : 0u, s_tmp, &tot);
        if (!FILL) { if (threadIdx.x == 0) byteSums[f * nb + blockIdx.x] = tot; }
        else off[f] = byteSums[f * nb + blockIdx.x] + e;
    }
    if (!FILL || !live) return;
    const uint8_t* const idxS[4] = { idx0, idx1, idx2, idx3 };
    const unsigned long long ib[4] = { ib0, ib1, ib2, ib3 };
    const uint8_t* const tblS[4] = { t0, t1, t2, t3 };
    const uint8_t* __restrict__ src = idxS[fmt] + ib[fmt] + off[fmt];
    const uint8_t* __restrict__ RGB = colors + ti * 6;
    const int len = 8 << fmt, pattern = (tile >> 6) & 255, orient = tile & 63;
    const uint8_t* __restrict__ lut = tblS[fmt] + ((size_t)pattern * 64 + orient) * len * 3;     // [(tile & 0x3FFF) * 3] << (3 + format), :330
    const bool inTable = pattern < nPat;
    const int diff[3] = { RGB[3] - RGB[0], RGB[4] - RGB[1], RGB[5] - RGB[2] };
    const int xCount = TX > 8 ? 2 : 1, lX = TX > 8 ? 8 : TX;
    int n = 0;
    for (int xa = 0; xa < xCount; xa++) for (int y = 0; y < TY; y++) for (int x = 0; x < lX; x++) {
        const int gx = x0 + x + xa * 8, gy = y0 + y;
        if (yk_dl_marked(tile4, stride4, gx, gy)) continue;
        const int e3 = src[n++];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int v = (inTable && e3 + c < len * 3) ? lut[e3 + c] : 251;    // beyond the loaded bank the reference reads its filler / unset memory
            planes[(size_t)c * planeSize + ((size_t)(gy >> 3) * tileW + (gx >> 3)) * 64 + (gy & 7) * 8 + (gx & 7)] = (uint8_t)(RGB[c] + ((diff[c] * v) >> 7));
        }
    }
}
// the cells of the pass's tiles are marked after the fill (a separate launch: the fill reads the mask of its own cells while it runs)
__global__ __launch_bounds__(256) void yk_dl_mark_kernel(const uint32_t* __restrict__ map, size_t nSlots, LutGeo g, int w, int h, uint32_t* __restrict__ tile4w, int stride4) {
    const size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    int x0, y0;
    if (!yk_dl_tile(map, nSlots, pos, g, w, h, x0, y0)) return;
    for (int cy = y0 >> 2; cy < (y0 + (1 << g.sy)) >> 2; cy++) for (int cx = x0 >> 2; cx < (x0 + (1 << g.sx)) >> 2; cx++) {
        const size_t byteI = (size_t)(cx >> 2) + (size_t)(cy >> 1) * stride4;
        const int bit = (((cx >> 1) & 1) << 2) | ((cy & 1) << 1) | (cx & 1);
        atomicOr(&tile4w[byteI >> 2], 1u << (bit + 8 * (byteI & 3)));
    }
}

void yk_lut_dec_destroy(yk_ctx* c) {
    if (!c->lutDec) return;
    for (auto& p : c->lutDec->tbl) if (p) (void)hipFree(p);
    delete c->lutDec; c->lutDec = nullptr;
}

extern "C" {

int yk_decode_assign_lut(yk_ctx* c, const uint8_t* lutFile, size_t lutBytes) {
    if (!c || !lutFile) return YK_ERR_BAD_ARG;
    if (lutBytes < 8 || lutFile[0] != 'L' || lutFile[1] != 'U' || lutFile[2] != 'L') return yk_fail(c, YK_ERR_BAD_ARG, "not a 3-D LUT file ('LUL')");       // YAIK_INVALID_LUT
    const int nPat = lutFile[5] + 1;
    if (lutBytes != 8 + (size_t)nPat * 3 * (64 + 32 + 16 + 8)) return yk_fail(c, YK_ERR_BAD_ARG, "LUT file size does not match its entry count");
    YK_HIP(c, hipSetDevice(c->device));
    yk_lut_dec_destroy(c);
    c->lutDec = new YkLutDecState(); c->lutDec->nPat = nPat;
    static const int axis[6][3] = { {0,1,2}, {0,2,1}, {1,0,2}, {1,2,0}, {2,0,1}, {2,1,0} };       // X, X[ZY], [YX]Z, YZX, ZXY, ZYX (:297-337)
    const uint8_t* stream = lutFile + 8;
    for (int bit = 3; bit <= 6; bit++) {
        const int len = 1 << bit;
        std::vector<uint8_t> T((size_t)nPat * 64 * len * 3, 251);              // slots 48..63 of every pattern stay filler (:400-404)
        for (int e = 0; e < nPat; e++) {
            const uint8_t* o[3] = { stream, stream + len, stream + 2 * len };
            uint8_t* fill = T.data() + (size_t)e * 64 * len * 3;
            for (int pat = 0; pat < 6; pat++) for (int flip = 0; flip < 8; flip++) for (int i = 0; i < len; i++) for (int k = 0; k < 3; k++) {
                const uint8_t v = o[axis[pat][k]][i];
                *fill++ = ((flip >> k) & 1) ? (uint8_t)(128 - v) : v;
            }
            stream += len * 3;
        }
        YK_HIP(c, hipMalloc(&c->lutDec->tbl[bit - 3], T.size() + 16));
        YK_HIP(c, hipMemcpy(c->lutDec->tbl[bit - 3], T.data(), T.size(), hipMemcpyHostToDevice));
    }
    return YK_OK;
}

int yk_decode_lut3d(yk_ctx* c, const uint8_t* const maps[6], const size_t mapBytes[6], const uint16_t* tiles, size_t nTiles, const uint8_t* colors,
                    const uint8_t* const idx[4], const size_t idxBytes[4], size_t consumed[6]) {
    if (!c || !maps || !mapBytes || !idx || !idxBytes) return YK_ERR_BAD_ARG;
    if (!c->dPlanes) return yk_fail(c, YK_ERR_STATE, "yk_decode_begin first");
    if (!c->lutDec) return yk_fail(c, YK_ERR_STATE, "yk_decode_assign_lut first");
    if (c->dSplit) return yk_fail(c, YK_ERR_STATE, "a '3DTL' chunk comes before the masks are split ('1DTL', plane-subset chunks)");
    YK_HIP(c, hipSetDevice(c->device));
    const int w = c->dw, h = c->dh;
    static const int sz[6][2] = { {4,3}, {3,4}, {3,3}, {3,2}, {2,3}, {2,2} };
    uint16_t* dTiles = nullptr; uint8_t* dColors = nullptr; uint8_t* dIdx[4] = {}; uint32_t* dMap = nullptr; uint32_t* sums = nullptr;
    auto freeAll = [&]() { (void)hipFree(dTiles); (void)hipFree(dColors); for (auto p : dIdx) (void)hipFree(p); (void)hipFree(dMap); (void)hipFree(sums); };
    size_t maxMap = 0; for (int k = 0; k < 6; k++) maxMap = mapBytes[k] > maxMap ? mapBytes[k] : maxMap;
    const size_t maxSlots = maxMap * 8, maxNb = (maxSlots + 1023) / 1024 + 1;
    hipError_t e = hipMalloc(&dTiles, nTiles * 2 + 64);
    if (e == hipSuccess) e = hipMalloc(&dColors, nTiles * 6 + 64);
    for (int f = 0; f < 4 && e == hipSuccess; f++) e = hipMalloc(&dIdx[f], idxBytes[f] + 256);
    if (e == hipSuccess) e = hipMalloc(&dMap, maxMap + 64);
    if (e == hipSuccess) e = hipMalloc(&sums, (5 * maxNb + 16) * sizeof(uint32_t));
    if (e == hipSuccess && nTiles) e = hipMemcpyAsync(dTiles, tiles, nTiles * 2, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && nTiles) e = hipMemcpyAsync(dColors, colors, nTiles * 6, hipMemcpyHostToDevice, c->stream);
    for (int f = 0; f < 4 && e == hipSuccess; f++) {
        e = hipMemsetAsync(dIdx[f], 0, idxBytes[f] + 256, c->stream);
        if (e == hipSuccess && idxBytes[f]) e = hipMemcpyAsync(dIdx[f], idx[f], idxBytes[f], hipMemcpyHostToDevice, c->stream);
    }
    unsigned long long tileBase = 0, ib[4] = { 0, 0, 0, 0 };
    for (int k = 0; k < 6 && e == hipSuccess; k++) {
        if (!maps[k] || !mapBytes[k]) continue;
        const LutGeo g = yk_lut_geo(sz[k][0], sz[k][1], w);
        size_t nSlots = (size_t)g.xBB * ((h + g.bigY - 1) / g.bigY) * g.bitCount;
        if (nSlots > mapBytes[k] * 8) nSlots = mapBytes[k] * 8;
        const size_t nb = (nSlots + 1023) / 1024;
        e = hipMemsetAsync(dMap, 0, maxMap + 64, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dMap, maps[k], mapBytes[k], hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) break;
        uint32_t* rankBase = sums; uint32_t* byteSums = sums + maxNb; uint32_t* totals = sums + 5 * maxNb;
        hipLaunchKernelGGL(yk_dl_rank_kernel, dim3((unsigned)nb), dim3(1024), 0, c->stream, dMap, nSlots, g, w, h, rankBase);
        hipLaunchKernelGGL(yk_u32_scanblocks_kernel, dim3(1), dim3(1024), 0, c->stream, rankBase, (int)nb, totals);
#define YK_DL_ARGS dMap, nSlots, g, w, h, rankBase, nb, byteSums, dTiles, tileBase, (unsigned long long)nTiles, dColors, dIdx[0], dIdx[1], dIdx[2], dIdx[3], ib[0], ib[1], ib[2], ib[3], \
                   c->lutDec->tbl[0], c->lutDec->tbl[1], c->lutDec->tbl[2], c->lutDec->tbl[3], c->lutDec->nPat, c->dPlanes, c->dPlaneSize, w >> 3, c->dTile4, (w + 15) >> 4
        hipLaunchKernelGGL(yk_dl_tile_kernel<false>, dim3((unsigned)nb), dim3(1024), 0, c->stream, YK_DL_ARGS);
        for (int f = 0; f < 4; f++) hipLaunchKernelGGL(yk_u32_scanblocks_kernel, dim3(1), dim3(1024), 0, c->stream, byteSums + f * nb, (int)nb, totals + 1 + f);
        hipLaunchKernelGGL(yk_dl_tile_kernel<true>, dim3((unsigned)nb), dim3(1024), 0, c->stream, YK_DL_ARGS);
#undef YK_DL_ARGS
        hipLaunchKernelGGL(yk_dl_mark_kernel, dim3((unsigned)((nSlots + 255) / 256)), dim3(256), 0, c->stream, dMap, nSlots, g, w, h, reinterpret_cast<uint32_t*>(c->dTile4), (w + 15) >> 4);
        e = hipGetLastError();
        uint32_t tot[5] = {};
        if (e == hipSuccess) e = hipMemcpyAsync(tot, totals, sizeof tot, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        unsigned long long used = tot[0];
        if (tileBase + used > nTiles) used = nTiles - tileBase;
        tileBase += used;
        for (int f = 0; f < 4; f++) ib[f] += tot[1 + f];
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    freeAll();
    if (e != hipSuccess) return yk_fail(c, YK_ERR_HIP, "3-D LUT decode", e);
    for (int f = 0; f < 4; f++) if (ib[f] > idxBytes[f]) return yk_fail(c, YK_ERR_RANGE, "index stream shorter than the tile maps need");
    if (consumed) { consumed[0] = (size_t)tileBase * 2; consumed[1] = (size_t)tileBase * 6; for (int f = 0; f < 4; f++) consumed[2 + f] = (size_t)ib[f]; }
    return YK_OK;
}

}  // extern "C"
