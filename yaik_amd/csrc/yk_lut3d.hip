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
// squared distance is 3 * 63^2; exported for the parity tests, the search does not read it), nearest-entry tables u8[4][64^3] for 6 / 5 / 4 / 3
// bits, factor tables s16[4][3][64], and 48 x 8 transformed subset points (yk_lut_point_table, 3 KB) from which the scoring computes the
// distances with v_dot4_i32_i8 instead of reading them (DESIGN 3.8 has the measurements of both forms).
// Not on the timed path of bench.py.
#include "yk_common.h"
#include "yk_device.h"
#include <algorithm>
#include <type_traits>
#include <cstring>
#include <vector>

#define LUT_CUBE (64 * 64 * 64)
#define LUT_FACTOR 128                      // FACTOR, EncoderContext.cpp:22
#define LUT_MAXPAT 64

struct YkLutPattern { uint16_t* dist; uint32_t* pos; short4* fac; int count; };      // device pointers
// ptab[pair * 8 + j] = the j-th point of a pattern's 3-bit subset as one orientation sees it (yk_lut_point_table).  Orientations of a pattern that
// see the same SET of points score alike on every tile and the reference keeps the first of them (a strict `<`), so only the first of each
// such group is a pair: pairs patStart[k] .. patStart[k + 1] - 1 belong to pattern k in the order of their orientations pairMode[pair] (the
// cumulative permutations reach at most 28 distinct orientations, a colour ramp along the cube's diagonal has 8).
// pos[pattern * 64^3 + cell] = the cell's nearest entry at 6 | 5 << 8 | 4 << 16 | 3 << 24 bits (one gather serves the four depths: the gathers into
// these 1 MB tables are what the search waits for); fac[(pattern * 4 + depth) * 64 + entry] = the entry's three factors (x, y, z, 0).  One allocation
// per table for the whole bank, passed as kernel arguments: no pointer per pattern to fetch first.
struct YkLutBank { const uint2* ptab; const uint2* ptabM; const uint8_t* pairMode; const int* patStart; const uint32_t* pos; const short4* fac; int nPat, nPairs; };
struct YkLutState {
    YkLutPattern pat[LUT_MAXPAT]; int nPat = 0;
    uint2* ptab = nullptr;                  // [<= LUT_MAXPAT * 48 pairs][8], 192 KB
    uint2* ptabM = nullptr;                 // the same points as rows of the MFMA scoring's A operand (see yk_lut_search_kernel), 64 zero rows behind them
    uint8_t* pairMode = nullptr; int* patStart = nullptr;
    std::vector<uint2> hPtab, hPtabM; std::vector<uint8_t> hPairMode; std::vector<int> hPatStart = { 0 };      // host copies, grown by every yk_lut_load_pattern
    uint32_t* posAll = nullptr;             // [LUT_MAXPAT][64^3], 64 MB (allocated with the first pattern)
    short4* facAll = nullptr;               // [LUT_MAXPAT][4][64]
    bool started = false;
    // corr3D_* streams (StartCorrelationSearch :7316-7364), device
    uint16_t* tileType = nullptr; uint8_t* color = nullptr; uint8_t* idx[4] = {}; uint8_t* map[6] = {};
    size_t nType = 0, nColor = 0, nIdx[4] = {}, mapBytes[6] = {};
    size_t capTiles = 0, capPix = 0; int capW = 0, capH = 0;
    // per-pass scratch, sized for the 4x4 pass and kept across passes and searches
    struct LutSlot* slots = nullptr; uint8_t* slotIdx = nullptr; uint32_t* sums = nullptr; uint32_t* list = nullptr;
};

// the 48 orientations of EvaluatePoint3D: its axis swap of group n >> 3 is applied to the RUNNING x, y, z on every iteration of its loop
// over n (encoder/EncoderContext.h:629-686), so entry n is a cumulative permutation; table[n] = source axis of x | y << 2 | z << 4
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
__device__ __forceinline__ void yk_lut_swap(int mode, int& x, int& y, int& z) {            // swap3D (:5314-5354), not cumulative; branch-free: lanes of a wave differ
    // 0: (x,y,z)  1: (x,z,y)  2: (y,x,z)  3: (y,z,x)  4: (z,x,y)  5: (z,y,x)
    const int nx = (mode == 2 || mode == 3) ? y : (mode >= 4 ? z : x);
    const int ny = (mode == 1 || mode == 3) ? z : ((mode == 2 || mode == 4) ? x : y);
    const int nz = (mode == 1 || mode == 4) ? y : ((mode == 3 || mode == 5) ? x : z);
    x = nx; y = ny; z = nz;
}

// Set3DPointCloud's scan of the cube (:4786-4813): nearest point of every cell among every (1 << step)-th point, first minimum wins; the
// distance field is overwritten by every step, i.e. it ends as the distance to the nearest point of the 3-bit subset
__global__ __launch_bounds__(256) void yk_lut_build_kernel(const uint8_t* __restrict__ pts, int count, uint16_t* __restrict__ dist, uint32_t* __restrict__ pos) {
    __shared__ uint8_t s_pts[64 * 3];
    if (threadIdx.x < count * 3) s_pts[threadIdx.x] = pts[threadIdx.x];
    __syncthreads();
    const int i3 = blockIdx.x * blockDim.x + threadIdx.x;
    if (i3 >= LUT_CUBE) return;
    const int x = i3 & 63, y = (i3 >> 6) & 63, z = i3 >> 12;
    uint32_t entries = 0;                                                   // the cell's nearest entry at 6 / 5 / 4 / 3 bits, one byte each
#pragma unroll
    for (int step = 0; step < 4; step++) {
        int minDist = 999999999, best = 0;
        for (int p = 0; p < count; p += 1 << step) {
            const int dx = x - s_pts[p * 3], dy = y - s_pts[p * 3 + 1], dz = z - s_pts[p * 3 + 2];
            const int d = dx * dx + dy * dy + dz * dz;
            if (d < minDist) { minDist = d; best = p >> step; }
        }
        entries |= (uint32_t)best << (8 * step);
        if (step == 3) dist[i3] = (uint16_t)minDist;
    }
    pos[i3] = entries;
}

// The scoring of EvaluatePoint3D sums, over a tile's pixels, the distance field at the cell orientation m maps the pixel's cell c to:
// T_m(c) = (f1(c[ax]), f2(c[ay]), f4(c[az])) with the axis permutation (ax, ay, az) of yk_lut_perm_table and f_b(v) = (m & b) ? 63 - v : v.
// The field is the squared distance to the nearest point of the 3-bit subset (<= 8 points, see yk_lut_build_kernel), and
// |T_m(c) - p|^2 = |c - p'|^2 with p'[ax] = f1(p.x), p'[ay] = f2(p.y), p'[az] = f4(p.z): instead of 48 permuted look-ups into a 64^3 table per
// pixel the search evaluates min_j (|p'_j|^2 - 2 c.p'_j) + |c|^2 directly, one v_dot4_i32_i8 per (pixel, point).  Entry = { bytes (-2p'x, -2p'y,
// -2p'z, 0) as int8, |p'|^2 }; subsets of fewer than 8 points repeat point 0 (the minimum does not change).
static void yk_lut_point_table(const uint8_t* pts, int count, uint2 out[48 * 8]) {
    uint8_t perm[48]; yk_lut_perm_table(perm);
    for (int m = 0; m < 48; m++) {
        const int ax = perm[m] & 3, ay = (perm[m] >> 2) & 3, az = (perm[m] >> 4) & 3;
        for (int j = 0; j < 8; j++) {
            const int p = (j * 8 < count) ? j * 8 : 0;
            int q[3];
            q[ax] = (m & 1) ? 63 - pts[p * 3] : pts[p * 3];
            q[ay] = (m & 2) ? 63 - pts[p * 3 + 1] : pts[p * 3 + 1];
            q[az] = (m & 4) ? 63 - pts[p * 3 + 2] : pts[p * 3 + 2];
            out[m * 8 + j].x = (uint32_t)(uint8_t)(-2 * q[0]) | ((uint32_t)(uint8_t)(-2 * q[1]) << 8) | ((uint32_t)(uint8_t)(-2 * q[2]) << 16);
            out[m * 8 + j].y = (uint32_t)(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
        }
    }
}

// min over a pair's eight points of |p|^2 - 2 c.p for one pixel: eight v_dot4_i32_i8 (a.b + c on four signed bytes, the three-source form: the builtin
// __builtin_amdgcn_sdot4 is selected as v_dot4c behind a v_mov of c, two instructions per product) and the minimum tree in ONE asm statement.
// The result of a v_dot4 must not be read by the next instructions: hipcc pads one state behind an asm statement, and with one dot product
// per statement a build with fewer registers put the reader of the last product right behind it (one SALU instruction between) -- wrong minima
// on some tiles, found by the 64-pattern test case; one s_nop 1 behind every product cured it.  Here every product has at least three VALU
// instructions between itself and its reader, and what leaves the statement is the result of a v_min3.
__device__ __forceinline__ int yk_nearest8(int a, const uint4 (&q)[4]) {
    int out, t0, t1, t2, t3, t4, t5, t6, t7, m1, m2;
    asm("v_dot4_i32_i8 %1, %11, %12, %13\n\t"
        "v_dot4_i32_i8 %2, %11, %14, %15\n\t"
        "v_dot4_i32_i8 %3, %11, %16, %17\n\t"
        "v_dot4_i32_i8 %4, %11, %18, %19\n\t"
        "v_dot4_i32_i8 %5, %11, %20, %21\n\t"
        "v_dot4_i32_i8 %6, %11, %22, %23\n\t"
        "v_dot4_i32_i8 %7, %11, %24, %25\n\t"
        "v_dot4_i32_i8 %8, %11, %26, %27\n\t"
        "v_min3_i32 %9, %1, %2, %3\n\t"
        "v_min3_i32 %10, %4, %5, %6\n\t"
        "v_min_i32 %9, %9, %10\n\t"
        "v_min3_i32 %0, %7, %8, %9"
        : "=&v"(out), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7), "=&v"(m1), "=&v"(m2)
        : "v"(a), "v"(q[0].x), "v"(q[0].y), "v"(q[0].z), "v"(q[0].w), "v"(q[1].x), "v"(q[1].y), "v"(q[1].z), "v"(q[1].w),
          "v"(q[2].x), "v"(q[2].y), "v"(q[2].z), "v"(q[2].w), "v"(q[3].x), "v"(q[3].y), "v"(q[3].z), "v"(q[3].w));
    return out;
}

#ifndef YK_LUT_ABLATE
#define YK_LUT_ABLATE 0                     // timing experiments only (tools/prof_lut.sh): 1 = no scoring, 2 = no entry evaluation, 3 / 4 / 5 = leave after the box / the scoring / the entry evaluation (nothing is matched)
#endif
typedef unsigned short y_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t yk_us2_bits(y_us2 a) { return (uint32_t)a.x | ((uint32_t)a.y << 16); }
__device__ __forceinline__ y_us2 yk_us2_from(int b) { y_us2 r = { (unsigned short)((uint32_t)b & 0xFFFFu), (unsigned short)((uint32_t)b >> 16) }; return r; }

// |a - b| through the SAD unit (with a literal 0 addend the compiler expands __usad into min / max / sub)
__device__ __forceinline__ uint32_t yk_absdiff(uint32_t a, uint32_t b) { uint32_t r; asm("v_sad_u32 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b)); return r; }

// minimum of the wave's 64 values as a wave-uniform number: four DPP steps inside the rows of 16 lanes, the four row minima through readlane.
// Every lane of the wave must be active.
template <int CTRL> __device__ __forceinline__ uint32_t yk_dpp_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t yk_wave_min_u32(uint32_t x) {
    x = min(x, yk_dpp_u32<0xB1>(x));                                        // quad_perm [1,0,3,2]
    x = min(x, yk_dpp_u32<0x4E>(x));                                        // quad_perm [2,3,0,1]
    x = min(x, yk_dpp_u32<0x141>(x));                                       // row_half_mirror
    x = min(x, yk_dpp_u32<0x140>(x));                                       // row_mirror
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)x, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)x, 16);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)x, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)x, 48);
    return min(min(r0, r1), min(r2, r3));
}

// sums / packed 16-bit minima over rows of 16 lanes (every lane of the row ends with the row's value), then over a segment of ROWS consecutive
// rows: lanes that start a segment read the other rows' values through readlane-free xor shuffles (one or two ds_bpermute instead of six)
__device__ __forceinline__ int yk_row_sum(int x) {
    x += (int)yk_dpp_u32<0xB1>((uint32_t)x); x += (int)yk_dpp_u32<0x4E>((uint32_t)x); x += (int)yk_dpp_u32<0x141>((uint32_t)x); x += (int)yk_dpp_u32<0x140>((uint32_t)x);
    return x;
}
template <int SEG> __device__ __forceinline__ int yk_seg_sum(int x) {       // SEG = 16, 32 or 64 lanes
    x = yk_row_sum(x);
    if (SEG >= 32) x += __shfl_xor(x, 16);
    if (SEG >= 64) x += __shfl_xor(x, 32);
    return x;
}
typedef unsigned short yk_us2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t yk_pkmin_u16(uint32_t a, uint32_t b) {
    yk_us2v x = { (unsigned short)(a & 0xFFFFu), (unsigned short)(a >> 16) }, y = { (unsigned short)(b & 0xFFFFu), (unsigned short)(b >> 16) };
    const yk_us2v m = __builtin_elementwise_min(x, y);
    return (uint32_t)m.x | ((uint32_t)m.y << 16);
}
template <int SEG> __device__ __forceinline__ uint32_t yk_seg_pkmin_u16(uint32_t x) {
    x = yk_pkmin_u16(x, yk_dpp_u32<0xB1>(x)); x = yk_pkmin_u16(x, yk_dpp_u32<0x4E>(x)); x = yk_pkmin_u16(x, yk_dpp_u32<0x141>(x)); x = yk_pkmin_u16(x, yk_dpp_u32<0x140>(x));
    if (SEG >= 32) x = yk_pkmin_u16(x, (uint32_t)__shfl_xor((int)x, 16));
    if (SEG >= 64) x = yk_pkmin_u16(x, (uint32_t)__shfl_xor((int)x, 32));
    return x;
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

// ---- the pass's candidate tiles: whole tiles with at least one pixel that some plane has not covered yet.  A tile of at most 16 x 8 pixels
// lies inside one 16x16 macro-tile, so its cells are a mask of that macro-tile's coverage words.  The others get their "not found" here: the
// search kernel is launched for the candidates only (on the bench frame half of the 4.2 M 4x4 slots have nothing left to code, and an empty
// workgroup still costs its dispatch).  list[0] = count (zeroed by the caller), then {slot, x0 | y0 << 16} pairs from word 2 on in any order: the results go to per-slot records.
__global__ __launch_bounds__(1024) void yk_lut_list_kernel(const uint16_t* __restrict__ cov, size_t covStride, int mtW, LutGeo g, int w, int h, size_t nSlots,
                                                           LutSlot* __restrict__ slots, uint32_t* __restrict__ list) {
    __shared__ uint32_t s_tmp[32], s_base;
    // four consecutive slots per thread and ONE atomic per workgroup of 4096 slots: a single address sustains about 88 atomics per
    // microsecond, one per wave of 64 slots made this kernel 0.3 ms for the 4x4 pass
    const size_t pos0 = ((size_t)blockIdx.x * 1024 + threadIdx.x) * 4;
    const int TX = 1 << g.sx, TY = 1 << g.sy;
    uint32_t cand = 0, xy[4] = { 0, 0, 0, 0 };
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const size_t pos = pos0 + r;
        if (pos >= nSlots) break;
        const uint32_t blk = (uint32_t)(pos / (uint32_t)g.bitCount), tt = (uint32_t)(pos % (uint32_t)g.bitCount);
        const int x0 = (int)(blk % (uint32_t)g.xBB) * g.bigX + (int)(tt % (uint32_t)g.tilesPerRow) * TX;
        const int y0 = (int)(blk / (uint32_t)g.xBB) * g.bigY + (int)(tt / (uint32_t)g.tilesPerRow) * TY;
        bool c = false;
        if (x0 + TX <= w && y0 + TY <= h) {                                  // partial tiles are never tried (:6304, :6311)
            const size_t mt = (size_t)(y0 >> 4) * mtW + (x0 >> 4);
            const uint32_t row = ((1u << (TX >> 2)) - 1u) << ((x0 >> 2) & 3);              // the tile's cells in one row of the macro-tile
            uint32_t mask = 0;
            for (int q = 0; q < (TY >> 2); q++) mask |= row << ((((y0 >> 2) & 3) + q) * 4);
            c = (~((uint32_t)cov[mt] & cov[covStride + mt] & cov[2 * covStride + mt]) & mask) != 0u;
        }
        xy[r] = (uint32_t)x0 | ((uint32_t)y0 << 16);
        if (c) cand |= 1u << r; else slots[pos].found = 0;
    }
    uint32_t tot;
    const uint32_t ex = yk_block_exscan((uint32_t)__popc(cand), s_tmp, &tot);
    if (threadIdx.x == 0) s_base = tot ? atomicAdd(&list[0], tot) : 0u;
    __syncthreads();
    uint32_t o = 1u + s_base + ex;
#pragma unroll
    for (int r = 0; r < 4; r++) if ((cand >> r) & 1u) reinterpret_cast<uint2*>(list)[o++] = make_uint2((uint32_t)(pos0 + r), xy[r]);   // the search does not divide again
}

// ---- one workgroup per tile: 128 threads for the 128-pixel shapes, one wave for the others -------------------------------------------
// Thread t holds pixel t & (nPix - 1) in the order computeValues3D walks the tile (left 8 columns first for 16-wide tiles, :5856-5859); with
// fewer than 64 pixels the wave holds the tile 64 / nPix times and every copy evaluates its own share of the patterns.
// Byte selectors (v_perm_b32) of the 48 orientations, orientation = flips | swap << 3:  c[] takes a pixel's cell from its plain (bytes 0-2 of the
// second operand) and flipped (bytes 0-2 of the first) 6-bit coordinates -- flip by channel, then swap3D; b[] applies the inverse swap to three bytes.
// Selected per lane with nested ?: the swaps compiled to exec-mask branches (half of the entry evaluation's instructions); a selector is one LDS read.
struct YkLutSelTab { uint32_t c[48], b[48]; };
constexpr YkLutSelTab yk_lut_make_sel_tab() {
    constexpr int SRC[6][3] = { { 0, 1, 2 }, { 0, 2, 1 }, { 1, 0, 2 }, { 1, 2, 0 }, { 2, 0, 1 }, { 2, 1, 0 } };      // swap3D (:5314-5354): new (x, y, z) = old [SRC]
    YkLutSelTab T{};
    for (int m = 0; m < 48; m++) {
        const int sw = m >> 3, back = sw == 3 ? 4 : (sw == 4 ? 3 : sw);                                               // the inverse of swap 3 is swap 4; the others are involutions
        uint32_t c = 0x0C000000u, b = 0x0C000000u;
        for (int j = 0; j < 3; j++) {
            const int src = SRC[sw][j];
            c |= (uint32_t)(src + (((m >> src) & 1) ? 4 : 0)) << (8 * j);
            b |= (uint32_t)SRC[back][j] << (8 * j);
        }
        T.c[m] = c; T.b[m] = b;
    }
    return T;
}
__constant__ YkLutSelTab yk_lut_sel_tab = yk_lut_make_sel_tab();

template <int SX, int SY>
__global__ __launch_bounds__((1 << (SX + SY)) > 64 ? 128 : 64) __attribute__((amdgpu_waves_per_eu(8, 8))) void yk_lut_search_kernel(const int32_t* __restrict__ pR, const int32_t* __restrict__ pG, const int32_t* __restrict__ pB, int strideElems,
                                                            int w, int h, LutGeo g, const YkLutBank bank, uint32_t* __restrict__ covCh32, size_t covStride,
                                                            int mtW, LutSlot* __restrict__ slots, uint8_t* __restrict__ slotIdx, uint32_t* __restrict__ bitmap,
                                                            const uint32_t* __restrict__ list) {
    __shared__ __attribute__((aligned(16))) int s_cell[128];               // normalised 6-bit coordinates x | y << 8 | z << 16 of the live pixels, compacted
    __shared__ int s_box[6], s_n, s_csq, s_boxw[2][8];                     // s_boxw: per wave, packed minima of the box + live pixels
    __shared__ int s_best[4];                                               // pattern, orientation, bit mode, found
    __shared__ int s_nrc[3];                                                // (1 << 20) / d per channel
    __shared__ float s_rcf[3];                                              // RN(1 / d) per channel
    // sized by the bank at launch (a 64-pattern bank needs 17 KB, the usual handful 2 KB)
    extern __shared__ int s_dyn[];
    const int nPat = bank.nPat;
    int* const s_sum = s_dyn;                                               // [nPairs <= nPat * 48]: the scores
    int* const s_mode = s_sum + nPat * 48;                                  // [nPat]: first orientation with the smallest score
    int (*const s_part)[8] = reinterpret_cast<int (*)[8]>(s_mode + nPat);   // [2 waves][nPat][8]: absErr of 6,5,4,3 bit; pixels with error > 5 of 6,5,4,3 bit
    __shared__ uint2 s_sel[48];                                             // cell and back-swap selectors of every orientation
    constexpr int TX = 1 << SX, TY = 1 << SY, nPix = TX * TY, NT = nPix > 64 ? 128 : 64;             // the shape is a template parameter: shuffle widths, copies and loops are constants
    const int t = threadIdx.x;
    const uint2 cand = reinterpret_cast<const uint2*>(list)[1 + blockIdx.x];  // the candidate tiles of the pass with their origins (yk_lut_list_kernel)
    const uint32_t pos = cand.x;
    const int x0 = (int)(cand.y & 0xFFFFu), y0 = (int)(cand.y >> 16);       // whole tiles only: partial ones never enter the list (:6304, :6311)
    if (t == 0) s_csq = 0;
    // buildBBox3D (:132-193): a pixel is out when all three planes already cover it; the box spans the others
    const int tp = t & (nPix - 1);                                          // the thread's pixel; the first nPix threads are the tile, the others copies
    constexpr int nCopies = NT / nPix;
    const int copy = t / nPix;
    int v[3] = { 0, 0, 0 };
    bool liveP;
    {
        int px, py;
        if (TX == 16) { px = (tp & 7) + ((tp / (8 * TY)) << 3); py = (tp % (8 * TY)) >> 3; } else { px = tp % TX; py = tp / TX; }
        const int gx = x0 + px, gy = y0 + py;
        const size_t mt = (size_t)(gy >> 4) * mtW + (gx >> 4);
        const int cbit = ((gy >> 2) & 3) * 4 + ((gx >> 2) & 3);
        const uint16_t* cov = reinterpret_cast<const uint16_t*>(covCh32);
        liveP = !(((cov[mt] & cov[covStride + mt] & cov[2 * covStride + mt]) >> cbit) & 1);
        // samples are 0..255 by contract (framework.h:82, like every kernel of this library); the mask keeps the box, and with it every table
        // index below, inside its range whatever the planes hold
        if (liveP) { const size_t pi = (size_t)gy * strideElems + gx; v[0] = pR[pi] & 255; v[1] = pG[pi] & 255; v[2] = pB[pi] & 255; }
    }
    const bool live = liveP && t < nPix;
    const unsigned long long bal = __ballot(live);
    {   // the box through wave reductions on packed 16-bit minima (max = 255 - min of 255 - v): the first version's LDS atomics on seven shared
        // words were serialised lane by lane, 2.3 of a pass's 8 ms at 8192^2
        // DPP steps inside the rows of 16 lanes; only the rows that hold pixels are combined (the copies of a small tile contribute 0xFFFF)
        constexpr int boxSeg = nPix < 64 ? nPix : 64;
        const uint32_t a = yk_seg_pkmin_u16<boxSeg>(live ? ((uint32_t)v[0] | ((uint32_t)v[1] << 16)) : 0xFFFFFFFFu);
        const uint32_t b = yk_seg_pkmin_u16<boxSeg>(live ? ((uint32_t)v[2] | ((uint32_t)(255 - v[0]) << 16)) : 0xFFFFFFFFu);
        const uint32_t c = yk_seg_pkmin_u16<boxSeg>(live ? ((uint32_t)(255 - v[1]) | ((uint32_t)(255 - v[2]) << 16)) : 0xFFFFFFFFu);
        if ((t & 63) == 0) {
            int* o = s_boxw[t >> 6];
            o[0] = (int)(a & 0xFFFFu); o[1] = (int)(a >> 16); o[2] = (int)(b & 0xFFFFu); o[3] = (int)(b >> 16); o[4] = (int)(c & 0xFFFFu); o[5] = (int)(c >> 16); o[6] = __popcll(bal);
        }
    }
    __syncthreads();
    if (t < 3) {                                                            // fold the two waves' halves; empty = {9999 x3, -1 x3} as the reference leaves it
        const int ml = min(s_boxw[0][t], NT > 64 ? s_boxw[1][t] : 0xFFFF), mh = min(s_boxw[0][3 + t], NT > 64 ? s_boxw[1][3 + t] : 0xFFFF);
        const int bl = ml == 0xFFFF ? 9999 : ml, bh = mh == 0xFFFF ? -1 : 255 - mh, bd = bh - bl;
        s_box[t] = bl; s_box[3 + t] = bh;
        // the channel's two reciprocals, computed once per tile in these three lanes instead of per pixel (an integer division is ~35 instructions of
        // the wave whoever needs the result): (1 << 20) / d for the scoring coordinates, RN(1 / d) for the exact float division of the entry evaluation
        s_nrc[t] = bd > 0 ? (1 << 20) / bd : 0;
        s_rcf[t] = bd > 0 ? __fdiv_rn(1.0f, (float)bd) : 0.0f;
    }
    if (t == 6) s_n = s_boxw[0][6] + (NT > 64 ? s_boxw[1][6] : 0);
    __syncthreads();
    const int pixels = s_n;
    int lo[3], d[3];
#pragma unroll
    for (int c = 0; c < 3; c++) { lo[c] = s_box[c]; d[c] = s_box[3 + c] - s_box[c]; }
    const bool accept = pixels != 0 && (((d[0] == 0) && d[1] != 0 && d[2] != 0) || ((d[1] == 0) && d[0] != 0 && d[2] != 0) || ((d[2] == 0) && d[0] != 0 && d[1] != 0) ||
                                        (d[0] != 0 && d[1] != 0 && d[2] != 0));                 // :6322-6326
    if (!accept || nPat == 0) { if (t == 0) slots[pos].found = 0; return; }
#if YK_LUT_ABLATE == 3
    if (t == 0) slots[pos].found = 0;
    return;
#endif
    if (t < 48) s_sel[t] = make_uint2(yk_lut_sel_tab.c[t], yk_lut_sel_tab.b[t]);                 // accepted tiles only; read after the next barriers
    const int rank = __popcll(bal & ((1ULL << (t & 63)) - 1ULL)) + (t >= 64 ? s_boxw[0][6] : 0);  // position among the live pixels (thread order = the reference's walk)
    {   // coordinates for the scoring (:6389-6405): (int)(((v - lo) * ((1 << 20) / d)) / 2^20 * 63) in float
        int csq = 0;
        if (live) {
            int q[3];
#pragma unroll
            for (int c = 0; c < 3; c++) { const int n = s_nrc[c]; const float f = __fmul_rn((float)((v[c] - lo[c]) * n), 1.0f / 1048576.0f); q[c] = (int)__fmul_rn(f, 63.0f); }   // / 2^20 is exact either way
            s_cell[rank] = q[0] | (q[1] << 8) | (q[2] << 16);
            csq = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
        }
        csq = yk_seg_sum<(nPix < 64 ? nPix : 64)>(csq);                      // lanes past the tile's pixels hold 0
        if ((t & 63) == 0 && csq) atomicAdd(&s_csq, csq);
    }
    __syncthreads();
    // EvaluatePoint3D: sum of the distance field over the tile's pixels for every (pattern, orientation); lane = one (pattern, orientation) pair
    // with its eight transformed points in registers (the next round's are loaded under this round's arithmetic), the pixels come as LDS
    // broadcasts: 8 dot products + 4 min3 + 1 add per pixel and lane.
    const int nPairs = bank.nPairs;
    if constexpr (nPix >= 32) {
        // Tiles of 32 pixels and more: the (point, pixel) products on the matrix cores (a 4x4 tile would fill half of the 32 columns: it keeps the
        // v_dot4 path below).  v_mfma_i32_32x32x16_i8 with rows = the 32 points of four
        // pairs, columns = 32 pixels, K = (x, y, z, 127, 1) against (-2 px, -2 py, -2 pz, a, b), |p|^2 = 127 a + b: D[point][pixel] = |p|^2 - 2 c.p.
        // Rows are ordered so that a lane's sixteen results (rows 8 b + 4 h + r of column lane & 31, h = lane >> 5; layout checked on the part by
        // tools/ubench/mfma_i8_layout.hip) are ALL eight points of two pairs: pair 2 h + (b >> 1), point 4 (b & 1) + r -- the minimum over a pair's
        // points stays inside the lane (4 v_min3 / v_min per pair and 32 pixels instead of 12.5 VALU instructions per pixel and 64 pairs), lanes 0-31
        // accumulate pairs 0 and 1 of the group, lanes 32-63 pairs 2 and 3, and a 32-lane sum per pair closes the group.
        typedef int yk_v16i __attribute__((ext_vector_type(16)));
        constexpr int maxChunks = nPix / 32;
        const int csqAll = s_csq;
        const int l = t & 63, wv0 = __builtin_amdgcn_readfirstlane(t >> 6);
        long Bop[maxChunks];
#pragma unroll
        for (int ch = 0; ch < maxChunks; ch++) {
            const int pp = ch * 32 + (l & 31);
            Bop[ch] = (l < 32 && pp < pixels) ? (long)(((unsigned long long)1u << 32) | (unsigned long long)((uint32_t)s_cell[pp] | 0x7F000000u)) : 0L;   // a missing pixel scores 0 everywhere
        }
        const int nCh = (YK_LUT_ABLATE == 1) ? 0 : (pixels + 31) >> 5;
        const int nGroups = (nPairs + 3) >> 2;
        // row of this lane in a group of 32 table entries (4 pairs x 8 points): row i = 8 b + 4 h + r  ->  pair 2 h + (b >> 1), point 4 (b & 1) + r
        const int rowOff = ((((l >> 2) & 1) * 2 + ((l >> 4) & 1)) * 8) + (((l >> 3) & 1) * 4) + (l & 3);
        // no branch around the load (the compiler then counts the loads in flight exactly): lanes 32-63, whose half of K is not used, and groups past
        // the last one read the zero rows behind the table
        auto loadA = [&](const int G) -> long {
            const uint2 e = bank.ptabM[l < 32 ? min(G, nGroups) * 32 + rowOff : nPairs * 8];
            return (long)(((unsigned long long)e.y << 32) | (unsigned long long)e.x);
        };
        // Four groups (16 pairs) per step and wave.  Their table rows are loaded a step ahead into the other of two register sets (no copy at the end of
        // a step, so the loads stay in flight under it), and their eight per-lane sums are added over the 32 pixel lanes TOGETHER: three combine steps
        // (lanes l and l ^ 1, l ^ 2, l ^ 8 split the values between them, each keeping half and adding the partner's share) leave one register, two
        // plain steps (l ^ 4, l ^ 16) finish it -- 25 instructions for eight sums instead of 8 x 6.  Lane bits afterwards: bit 0 = second pair of the
        // lane half, bit 1 | bit 3 << 1 = group of the step, bit 5 = lane half; lanes with bits 2 and 4 clear store.
        constexpr int STEP = (NT >> 6) * 4;
        auto loadA4 = [&](const int g0, long (&A)[4]) {
#pragma unroll
            for (int g = 0; g < 4; g++) A[g] = loadA(g0 + g);
        };
        auto combine = [&](const bool low, const int a, const int b, auto ctrl) {
            const int keep = low ? a : b, send = low ? b : a;
            return keep + (int)yk_dpp_u32<decltype(ctrl)::value>((uint32_t)send);
        };
        auto process = [&](const int g0, const long (&A)[4]) {
            int vs[8];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                int accA = 0, accB = 0;
                if (g0 + g < nGroups) {
#pragma unroll
                    for (int ch = 0; ch < maxChunks; ch++) {
                        if (ch < nCh) {
                            const yk_v16i z = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
                            const yk_v16i dd = __builtin_amdgcn_mfma_i32_32x32x16_i8(A[g], Bop[ch], z, 0, 0, 0);
                            accA += min(min(min(dd[0], dd[1]), min(dd[2], dd[3])), min(min(dd[4], dd[5]), min(dd[6], dd[7])));
                            accB += min(min(min(dd[8], dd[9]), min(dd[10], dd[11])), min(min(dd[12], dd[13]), min(dd[14], dd[15])));
                        }
                    }
                }
                vs[2 * g] = accA; vs[2 * g + 1] = accB;
            }
            const bool low1 = !(l & 1), low2 = !(l & 2), low3 = !(l & 8);
            const std::integral_constant<int, 0xB1> X1; const std::integral_constant<int, 0x4E> X2; const std::integral_constant<int, 0x128> X8;   // quad_perm [1,0,3,2], [2,3,0,1], row_ror:8
            const int w0 = combine(low1, vs[0], vs[1], X1), w1 = combine(low1, vs[2], vs[3], X1), w2 = combine(low1, vs[4], vs[5], X1), w3 = combine(low1, vs[6], vs[7], X1);
            const int x0 = combine(low2, w0, w1, X2), x1 = combine(low2, w2, w3, X2);
            int y = combine(low3, x0, x1, X8);
            y += (int)yk_dpp_u32<0x1B>(yk_dpp_u32<0x141>((uint32_t)y));     // lane l ^ 4: row_half_mirror, then the quad reversed
            y += __shfl_xor(y, 16);
            if (!(l & 20)) {
                const int pi = (g0 + ((l >> 2) & 2) + ((l >> 1) & 1)) * 4 + (l >> 5) * 2 + (l & 1);
                if (pi < nPairs) s_sum[pi] = y + csqAll;
            }
        };
        long A0[4], A1[4];
        int g0 = wv0 * 4;
        loadA4(g0, A0);
        while (g0 < nGroups) {
            loadA4(g0 + STEP, A1);
            process(g0, A0);
            g0 += STEP;
            if (g0 >= nGroups) break;
            loadA4(g0 + STEP, A0);
            process(g0, A1);
            g0 += STEP;
        }
    } else {
        const int csqAll = s_csq;
        auto points = [&](const int pi, uint4 (&q)[4]) {
            const uint4* __restrict__ pt = reinterpret_cast<const uint4*>(bank.ptab + (size_t)min(pi, nPairs - 1) * 8);
            q[0] = pt[0]; q[1] = pt[1]; q[2] = pt[2]; q[3] = pt[3];
        };
        uint4 q[4];
        points(t, q);
        for (int pi0 = t & ~63; pi0 < nPairs; pi0 += NT) {                  // wave-uniform bound
            const int pi = pi0 + (t & 63);
            uint4 qn[4];
            const bool more = pi0 + NT < nPairs;
            if (more) points(pi + NT, qn);
            auto nearest = [&](const int a) { return yk_nearest8(a, q); };
            int sum = 0, p0 = 0;
#if YK_LUT_ABLATE == 1
            p0 = pixels;
#endif
            for (; p0 + 4 <= pixels; p0 += 4) {
                const int4 a = *reinterpret_cast<const int4*>(&s_cell[p0]);
                sum += nearest(a.x) + nearest(a.y) + nearest(a.z) + nearest(a.w);
            }
            for (; p0 < pixels; p0++) sum += nearest(s_cell[p0]);
            if (pi < nPairs) s_sum[pi] = sum + csqAll;
            if (more) {
#pragma unroll
                for (int i = 0; i < 4; i++) q[i] = qn[i];
            }
        }
    }
    __syncthreads();
#if YK_LUT_ABLATE == 4
    if (t == 0) slots[pos].found = s_sum[0] == 12345;
    return;
#endif
    // GetEvaluation3D (:697-711): first minimum of sum / (samples * 1024.0f) in float.  The sums are integers below 2^21 (128 pixels x 3 x 63^2) and the
    // divisor is the same for all of them, so two different sums differ by more than 2^-21 relatively and their float quotients differ too: the first
    // minimum of the quotients is the first minimum of the integers.  A wave per pattern: key = sum << 6 | lane, one wave minimum gives both.
    // Two patterns per step, one in each half of the wave (a pattern has at most 28 distinct orientations, see YkLutBank; one with more than 32
    // pairs would take the whole wave).
    {
        const int l = t & 63, wv0 = __builtin_amdgcn_readfirstlane(t >> 6);
        for (int k = wv0 * 2; k < nPat; k += (NT >> 6) * 2) {
            const int f0 = bank.patStart[k], f1 = bank.patStart[k + 1], f2 = k + 1 < nPat ? bank.patStart[k + 2] : f1;
            const int cnt0 = f1 - f0, cnt1 = f2 - f1;
            if (cnt0 <= 32 && cnt1 <= 32) {
                const int h = l >> 5, lh = l & 31, first = h ? f1 : f0, cnt = h ? cnt1 : cnt0;
                const uint32_t pm = lh < cnt ? bank.pairMode[first + lh] : 0u;  // in flight under the reduction, picked by lane number after it
                uint32_t key = lh < cnt ? (((uint32_t)s_sum[first + lh] << 6) | (uint32_t)lh) : 0xFFFFFFFFu;
                key = min(key, yk_dpp_u32<0xB1>(key)); key = min(key, yk_dpp_u32<0x4E>(key)); key = min(key, yk_dpp_u32<0x141>(key)); key = min(key, yk_dpp_u32<0x140>(key));
                const uint32_t mn0 = min((uint32_t)__builtin_amdgcn_readlane((int)key, 0), (uint32_t)__builtin_amdgcn_readlane((int)key, 16));
                const uint32_t mn1 = min((uint32_t)__builtin_amdgcn_readlane((int)key, 32), (uint32_t)__builtin_amdgcn_readlane((int)key, 48));
                const int mode0 = __builtin_amdgcn_readlane((int)pm, (int)(mn0 & 31u)), mode1 = __builtin_amdgcn_readlane((int)pm, (int)(32u + (mn1 & 31u)));
                if (l == 0) { s_mode[k] = mode0; if (cnt1) s_mode[k + 1] = mode1; }
            } else {
                for (int kk = k; kk < min(k + 2, nPat); kk++) {
                    const int first0 = bank.patStart[kk], cnt = bank.patStart[kk + 1] - first0;
                    const uint32_t pm = l < cnt ? bank.pairMode[first0 + l] : 0u;
                    const uint32_t key = l < cnt ? (((uint32_t)s_sum[first0 + l] << 6) | (uint32_t)l) : 0xFFFFFFFFu;
                    const uint32_t mn = yk_wave_min_u32(key);
                    const int mode = __builtin_amdgcn_readlane((int)pm, (int)(mn & 63u));
                    if (l == 0) s_mode[kk] = mode;
                }
            }
        }
    }
    __syncthreads();
    // computeValues3D for every pattern at its best orientation: per pixel the entry at 6 / 5 / 4 / 3 bits and its worst channel error.
    // Its early exit (all four depths rejected at the end of a row) returns what the full pass returns, so the sums are order-free.
    // the pixel's position in the box (:5871-5887), the same for every pattern and depth; lanes without a pixel sit on the box's low corner
    float relp[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float rel = liveP ? (float)(v[c] - lo[c]) : 0.0f;
        rel = yk_div_exact(rel, (float)d[c], s_rcf[c]);                    // == __fdiv_rn for 0..255 / 1..255 (yk_selftest 0); d == 0: rel is 0 and stays 0
        relp[c] = __fmul_rn(rel, 63.0f);
    }
    // plain and flipped 6-bit coordinates, W = v - lo and d as packed bytes: an orientation's flip + swap is one v_perm_b32 with its selector
    const uint32_t MI = (uint32_t)(int)relp[0] | ((uint32_t)(int)relp[1] << 8) | ((uint32_t)(int)relp[2] << 16);
    const uint32_t MF = (uint32_t)(int)__fsub_rn(63.0f, relp[0]) | ((uint32_t)(int)__fsub_rn(63.0f, relp[1]) << 8) | ((uint32_t)(int)__fsub_rn(63.0f, relp[2]) << 16);
    const uint32_t Wp = liveP ? ((uint32_t)(v[0] - lo[0]) | ((uint32_t)(v[1] - lo[1]) << 8) | ((uint32_t)(v[2] - lo[2]) << 16)) : 0u;
    const uint32_t Dp = (uint32_t)d[0] | ((uint32_t)d[1] << 8) | ((uint32_t)d[2] << 16);                // 0..255 each once a tile is accepted
    // the cell of this pixel in pattern space under orientation `mode`
    auto cellOf = [&](int mode) {
        const uint32_t cm = __builtin_amdgcn_perm(MF, MI, s_sel[mode].x);
        return (int)((cm & 63u) | ((cm >> 2) & 0xFC0u) | ((cm >> 4) & 0x3F000u));
    };
    // per pattern: the four depths' worst-channel errors of this pixel, summed over the tile's lanes with shuffles that stay inside the copy
    // (two 16-bit sums per word: a tile holds at most 128 pixels of error <= 255) and ballots for the ">5" counts; LDS atomics on eight
    // shared words per pattern serialised all 64 lanes of a wave and were the larger half of the first version's time
    {
        const int wv = t >> 6;
        constexpr int seg = nPix < 64 ? nPix : 64;
        const unsigned long long segMask = seg == 64 ? ~0ULL : (((1ULL << seg) - 1ULL) << ((t & 63) - tp));
        // The colour of an entry (:5889-5925) is lo + ((swap(flip(factors)) * d) / FACTOR) per channel and its error max |colour - v|.  Seen from the
        // pattern's axes: axis a drives the channel the swap sends it to, so with W = v - lo and d permuted the other way ONCE per pattern, an axis
        // costs mad (flip folded in: (F ? 128 - f : f) * d = f * (+-d) + (F ? 128 d : 0)), shift (all operands are >= 0), |x - W| per depth.
        auto evalPattern = [&](const int k, int (&w4)[4]) {
            const int mode = s_mode[k];
            const uint2 sel = s_sel[mode];
            const uint32_t cm = __builtin_amdgcn_perm(MF, MI, sel.x);
            const uint32_t entries = bank.pos[(size_t)k * LUT_CUBE + ((cm & 63u) | ((cm >> 2) & 0xFC0u) | ((cm >> 4) & 0x3F000u))];
            const uint32_t Wq = __builtin_amdgcn_perm(0u, Wp, sel.y), Dq = __builtin_amdgcn_perm(0u, Dp, sel.y);
            int W[3], SD[3], BD[3];
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const int Da = (int)((Dq >> (8 * a)) & 255u);
                const bool F = (mode >> a) & 1;
                W[a] = (int)((Wq >> (8 * a)) & 255u); SD[a] = F ? -Da : Da; BD[a] = F ? LUT_FACTOR * Da : 0;
            }
#pragma unroll
            for (int depth = 0; depth < 4; depth++) {
                const short4 f = bank.fac[(k * 4 + depth) * 64 + (int)((entries >> (8 * depth)) & 255u)];
                const uint32_t e0 = yk_absdiff((uint32_t)(f.x * SD[0] + BD[0]) >> 7, (uint32_t)W[0]);
                const uint32_t e1 = yk_absdiff((uint32_t)(f.y * SD[1] + BD[1]) >> 7, (uint32_t)W[1]);
                const uint32_t e2 = yk_absdiff((uint32_t)(f.z * SD[2] + BD[2]) >> 7, (uint32_t)W[2]);
                w4[depth] = liveP ? (int)max(max(e0, e1), e2) : 0;
            }
        };
        auto reducePattern = [&](const int k, const int (&w4)[4]) {
            int s01 = w4[0] | (w4[1] << 16), s23 = w4[2] | (w4[3] << 16);
            s01 = yk_seg_sum<seg>(s01); s23 = yk_seg_sum<seg>(s23);
            const int c0 = __popcll(__ballot(w4[0] > 5) & segMask), c1 = __popcll(__ballot(w4[1] > 5) & segMask);
            const int c2 = __popcll(__ballot(w4[2] > 5) & segMask), c3 = __popcll(__ballot(w4[3] > 5) & segMask);
            if ((tp & 63) == 0 && k < nPat) {
                int* a = s_part[wv * nPat + k];
                a[0] = s01 & 0xFFFF; a[1] = s01 >> 16; a[2] = s23 & 0xFFFF; a[3] = s23 >> 16; a[4] = c0; a[5] = c1; a[6] = c2; a[7] = c3;
            }
        };
        // two patterns per copy and round: their eight look-up chains (cell -> entry -> three factors) are in flight together.  Lanes without a
        // pixel and copies without a pattern run the same loads on a harmless cell / the last pattern (every lane of a wave takes part in the
        // reductions) and contribute nothing.
        for (int kk = 0; kk < (YK_LUT_ABLATE == 2 ? 0 : nPat); kk += 2 * nCopies) {
            const int ka = kk + copy, kb = ka + nCopies;
            const bool second = kk + nCopies < nPat;
            int wa[4], wb[4] = { 0, 0, 0, 0 };
            evalPattern(min(ka, nPat - 1), wa);
            if (second) evalPattern(min(kb, nPat - 1), wb);
            reducePattern(ka, wa);
            if (second) reducePattern(kb, wb);
        }
    }
    __syncthreads();
#if YK_LUT_ABLATE == 5
    if (t == 0) slots[pos].found = s_part[0][4] == 12345;
    return;
#endif
    if (t < 64) {
        // :6066-6069 (lowest depth that is not rejected, a depth is rejected when more than 3 pixels are off by more than 5) and the choice
        // among patterns :6486 (smallest summed error, the LATER pattern on a tie): lane = pattern, key = error << 14 | (4095 - pattern) << 2 | depth,
        // one wave minimum instead of a loop over the patterns in one lane
        uint32_t bestKey = 0xFFFFFFFFu;
        for (int k = t; k < nPat; k += 64) {
            int acc[8];
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = s_part[k][i] + (NT > 64 ? s_part[nPat + k][i] : 0);
            int res = 4, diff = 0;
            if (acc[4] <= 3) { diff = acc[0]; res = 3; }
            if (acc[5] <= 3) { diff = acc[1]; res = 2; }
            if (acc[6] <= 3) { diff = acc[2]; res = 1; }
            if (acc[7] <= 3) { diff = acc[3]; res = 0; }
            if (res != 4) bestKey = min(bestKey, ((uint32_t)diff << 14) | ((uint32_t)(4095 - k) << 2) | (uint32_t)res);
        }
        const uint32_t mk = yk_wave_min_u32(bestKey);
        if (t == 0) {
            const int found = mk != 0xFFFFFFFFu, bestK = found ? 4095 - (int)((mk >> 2) & 4095u) : -1, bestMode = found ? (int)(mk & 3u) : 4;
            const int bestOrient = found ? s_mode[bestK] : 0;
            s_best[0] = bestK; s_best[1] = bestOrient; s_best[2] = bestMode; s_best[3] = found;
            LutSlot sl;
            sl.found = (uint8_t)found; sl.pixels = (uint8_t)pixels; sl.mode = (uint8_t)bestMode; sl.pad = 0;
            sl.type = (uint16_t)(bestOrient | (bestMode << 14) | ((found ? bestK : 0) << 6));        // :6559
            for (int c = 0; c < 6; c++) sl.box[c] = (uint8_t)s_box[c];
            slots[pos] = sl;
            if (found) atomicOr(&bitmap[pos >> 5], 1u << (pos & 31));
        }
    }
    __syncthreads();
    if (!s_best[3]) return;
    // the winner's indices in stream order
    if (live) {
        const uint32_t entries = bank.pos[(size_t)s_best[0] * LUT_CUBE + cellOf(s_best[1])];
        slotIdx[(size_t)pos * nPix + rank] = (uint8_t)(entries >> (8 * (3 - s_best[2])));
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
    for (int k = 0; k < S->nPat; k++) F(S->pat[k].dist);
    F(S->ptab); F(S->ptabM); F(S->pairMode); F(S->patStart); F(S->posAll); F(S->facAll); F(S->slots); F(S->slotIdx); F(S->sums); F(S->list); F(S->tileType); F(S->color);
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
    if (!c->lut) c->lut = new YkLutState();
    YkLutState* S = c->lut;
    if (!S->ptab) YK_HIP(c, hipMalloc(&S->ptab, (size_t)LUT_MAXPAT * 48 * 8 * sizeof(uint2)));
    if (!S->ptabM) YK_HIP(c, hipMalloc(&S->ptabM, ((size_t)LUT_MAXPAT * 48 * 8 + 64) * sizeof(uint2)));
    if (!S->pairMode) YK_HIP(c, hipMalloc(&S->pairMode, (size_t)LUT_MAXPAT * 48));
    if (!S->patStart) YK_HIP(c, hipMalloc(&S->patStart, (LUT_MAXPAT + 1) * sizeof(int)));
    if (!S->posAll) YK_HIP(c, hipMalloc(&S->posAll, (size_t)LUT_MAXPAT * LUT_CUBE * sizeof(uint32_t)));
    if (!S->facAll) YK_HIP(c, hipMalloc(&S->facAll, (size_t)LUT_MAXPAT * 4 * 64 * sizeof(short4)));
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
    P.pos = S->posAll + (size_t)S->nPat * LUT_CUBE;
    P.fac = S->facAll + (size_t)S->nPat * 4 * 64;
    short4 fac4[4][64];
    for (int step = 0; step < 4; step++)
        for (int e = 0; e < 64; e++) fac4[step][e] = make_short4(fac[step][0][e], fac[step][1][e], fac[step][2][e], 0);
    uint8_t* dPts = nullptr;
    YK_HIP(c, hipMalloc(&dPts, 64 * 3));
    YK_HIP(c, hipMemcpyAsync(dPts, pts, (size_t)count * 3, hipMemcpyHostToDevice, c->stream));
    YK_HIP(c, hipMemcpyAsync(P.fac, fac4, sizeof fac4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(yk_lut_build_kernel, dim3(LUT_CUBE / 256), dim3(256), 0, c->stream, dPts, count, P.dist, P.pos);
    {
        uint2 ptab[48 * 8];
        yk_lut_point_table(pts, count, ptab);
        const size_t firstPair = S->hPairMode.size();
        for (int m = 0; m < 48; m++) {
            uint2* e = ptab + m * 8;                                            // as a set: the order of the points does not matter to a minimum
            std::sort(e, e + 8, [](const uint2& a, const uint2& b) { return a.x != b.x ? a.x < b.x : a.y < b.y; });
            bool seen = false;
            for (size_t q = firstPair; q < S->hPairMode.size() && !seen; q++) seen = memcmp(&S->hPtab[q * 8], e, 8 * sizeof(uint2)) == 0;
            if (!seen) { S->hPtab.insert(S->hPtab.end(), e, e + 8); S->hPairMode.push_back((uint8_t)m); }
        }
        S->hPatStart.push_back((int)S->hPairMode.size());
        YK_HIP(c, hipMemcpyAsync(S->ptab, S->hPtab.data(), S->hPtab.size() * sizeof(uint2), hipMemcpyHostToDevice, c->stream));
        // MFMA rows: bytes (-2x, -2y, -2z, a, b, 0, 0, 0) with |p|^2 = 127 a + b (both fit a signed byte); 64 zero rows behind the last pair
        // (a short last group, groups past the end of a four-group step and the lanes of the unused half of K read them)
        S->hPtabM.assign(S->hPtab.size() + 64, make_uint2(0u, 0u));
        for (size_t q = 0; q < S->hPtab.size(); q++) S->hPtabM[q] = make_uint2((S->hPtab[q].x & 0x00FFFFFFu) | ((S->hPtab[q].y / 127u) << 24), S->hPtab[q].y % 127u);
        YK_HIP(c, hipMemcpyAsync(S->ptabM, S->hPtabM.data(), S->hPtabM.size() * sizeof(uint2), hipMemcpyHostToDevice, c->stream));
        YK_HIP(c, hipMemcpyAsync(S->pairMode, S->hPairMode.data(), S->hPairMode.size(), hipMemcpyHostToDevice, c->stream));
        YK_HIP(c, hipMemcpyAsync(S->patStart, S->hPatStart.data(), S->hPatStart.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    }
    YK_HIP(c, hipGetLastError());
    YK_HIP(c, hipStreamSynchronize(c->stream));
    (void)hipFree(dPts);
    if (index) *index = S->nPat;
    S->nPat++;
    return YK_OK;
}

int yk_lut_pattern_tables(yk_ctx* c, int pattern, int16_t* factors /*4*3*64*/, uint16_t* distanceField /*64^3*/, uint8_t* positions /*4*64^3*/) {
    if (!c || !c->lut || pattern < 0 || pattern >= c->lut->nPat) return YK_ERR_BAD_ARG;
    YK_HIP(c, hipSetDevice(c->device));
    const YkLutPattern& P = c->lut->pat[pattern];
    if (factors) {                                                             // exported in the reference's layout [depth][channel][entry]
        short4 fac4[4][64];
        YK_HIP(c, hipMemcpy(fac4, P.fac, sizeof fac4, hipMemcpyDeviceToHost));
        for (int step = 0; step < 4; step++)
            for (int e = 0; e < 64; e++) { factors[(step * 3 + 0) * 64 + e] = fac4[step][e].x; factors[(step * 3 + 1) * 64 + e] = fac4[step][e].y; factors[(step * 3 + 2) * 64 + e] = fac4[step][e].z; }
    }
    if (distanceField) YK_HIP(c, hipMemcpy(distanceField, P.dist, LUT_CUBE * sizeof(uint16_t), hipMemcpyDeviceToHost));
    if (positions) {                                                           // [depth][cell]
        std::vector<uint32_t> e(LUT_CUBE);
        YK_HIP(c, hipMemcpy(e.data(), P.pos, LUT_CUBE * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (int step = 0; step < 4; step++)
            for (int i = 0; i < LUT_CUBE; i++) positions[(size_t)step * LUT_CUBE + i] = (uint8_t)(e[i] >> (8 * step));
    }
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
    static const int sz[6][2] = { {4,3}, {3,4}, {3,3}, {3,2}, {2,3}, {2,2} };
    const size_t capTiles = (size_t)(w / 4) * (h / 4) + 16, capPix = (size_t)w * h + 128;
    if (w != S->capW || h != S->capH) {                      // a new image size: the streams, the maps and the per-pass scratch
        auto F = [](auto*& p) { if (p) { (void)hipFree((void*)p); p = nullptr; } };
        YK_HIP(c, hipStreamSynchronize(c->stream));
        F(S->tileType); F(S->color); for (auto& p : S->idx) F(p); for (auto& p : S->map) F(p); F(S->slots); F(S->slotIdx); F(S->sums); F(S->list);
        S->capTiles = S->capPix = 0; S->capW = S->capH = 0;
        YK_HIP(c, hipMalloc(&S->tileType, capTiles * 2));
        YK_HIP(c, hipMalloc(&S->color, capTiles * 6));
        for (auto& p : S->idx) YK_HIP(c, hipMalloc(&p, capPix));
        size_t maxSlots = 0;
        for (int k = 0; k < 6; k++) {
            const LutGeo g = yk_lut_geo(sz[k][0], sz[k][1], w);
            S->mapBytes[k] = (size_t)g.xBB * ((h + g.bigY - 1) / g.bigY) * g.bitCount;       // BitmapSwizzleMapSize (:7310): a bit count used as the byte size
            YK_HIP(c, hipMalloc(&S->map[k], S->mapBytes[k] + 16));
            maxSlots = std::max(maxSlots, S->mapBytes[k]);                                    // = the pass's tile slots
        }
        YK_HIP(c, hipMalloc(&S->slots, maxSlots * sizeof(LutSlot)));
        YK_HIP(c, hipMalloc(&S->slotIdx, (size_t)(w + 64) * (h + 64)));                      // slots x pixels per tile = whole swizzle blocks (64 x 64 at most), any shape
        YK_HIP(c, hipMalloc(&S->sums, (5 * ((maxSlots + 1023) / 1024) + 16) * sizeof(uint32_t)));
        YK_HIP(c, hipMalloc(&S->list, (maxSlots + 1) * sizeof(uint2)));
        S->capTiles = capTiles; S->capPix = capPix; S->capW = w; S->capH = h;
    }
    for (int k = 0; k < 6; k++) YK_HIP(c, hipMemsetAsync(S->map[k], 0, S->mapBytes[k] + 16, c->stream));
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
    LutSlot* const slots = S->slots; uint8_t* const slotIdx = S->slotIdx; uint32_t* const sums = S->sums;    // yk_lut_start sized them
    YkLutBank bank; bank.ptab = S->ptab; bank.ptabM = S->ptabM; bank.pairMode = S->pairMode; bank.patStart = S->patStart; bank.pos = S->posAll; bank.fac = S->facAll;
    bank.nPat = S->nPat; bank.nPairs = (int)S->hPairMode.size();
    uint32_t nCand = 0;
    YK_HIP(c, hipMemsetAsync(S->list, 0, sizeof(uint32_t), c->stream));
    { int rc = yk_stage_begin(c, YK_STAGE_LUT3D); if (rc) return rc; }
    hipLaunchKernelGGL(yk_lut_list_kernel, dim3((unsigned)((nSlots + 4095) / 4096)), dim3(1024), 0, c->stream, c->covCh, c->covChStride, c->mtW, g, w, h, nSlots, slots, S->list);
    { int rc = yk_stage_end(c, YK_STAGE_LUT3D); if (rc) return rc; }
    YK_HIP(c, hipGetLastError());
    YK_HIP(c, hipMemcpyAsync(&nCand, S->list, sizeof nCand, hipMemcpyDeviceToHost, c->stream));
    YK_HIP(c, hipStreamSynchronize(c->stream));                              // the grid of the search is the number of candidates
    if (nCand) {
        { int rc = yk_stage_begin(c, YK_STAGE_LUT3D); if (rc) return rc; }
#define YK_LUT_LAUNCH(SX_, SY_) hipLaunchKernelGGL((yk_lut_search_kernel<SX_, SY_>), dim3(nCand), dim3(nPix > 64 ? 128 : 64), (size_t)S->nPat * (48 + 1 + 16) * sizeof(int), c->stream,       \
                           c->plane[0], c->plane[1], c->plane[2], c->strideElems, w, h, g, bank, reinterpret_cast<uint32_t*>(c->covCh), c->covChStride, c->mtW, slots, slotIdx,               \
                           reinterpret_cast<uint32_t*>(S->map[g.mapId]), S->list)
        switch (g.mapId) {                                                  // one instantiation per tile shape
            case 0: YK_LUT_LAUNCH(4, 3); break;
            case 1: YK_LUT_LAUNCH(3, 4); break;
            case 2: YK_LUT_LAUNCH(3, 3); break;
            case 3: YK_LUT_LAUNCH(3, 2); break;
            case 4: YK_LUT_LAUNCH(2, 3); break;
            default: YK_LUT_LAUNCH(2, 2); break;
        }
#undef YK_LUT_LAUNCH
        { int rc = yk_stage_end(c, YK_STAGE_LUT3D); if (rc) return rc; }
    }
    hipLaunchKernelGGL(yk_lut_count_kernel, dim3((unsigned)nb), dim3(1024), 0, c->stream, slots, nSlots, sums, nb);
    hipLaunchKernelGGL(yk_lut_scan_kernel, dim3(1), dim3(1024), 0, c->stream, sums, nb, sums + 5 * nb);
    LutStreams out; out.tileType = S->tileType; out.color = S->color; out.nType = S->nType; out.nColor = S->nColor;
    for (int m = 0; m < 4; m++) { out.idx[m] = S->idx[m]; out.nIdx[m] = S->nIdx[m]; }
    hipLaunchKernelGGL(yk_lut_emit_kernel, dim3((unsigned)nb), dim3(1024), 0, c->stream, slots, slotIdx, nPix, nSlots, sums, nb, out);
    hipError_t e = hipGetLastError();
    uint32_t tot[5] = {};
    if (e == hipSuccess) e = hipMemcpyAsync(tot, sums + 5 * nb, sizeof tot, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
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
    bool shortStream = false;
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
        // The counting launch and the scans know what this pass would consume: tiles / colours and index bytes per depth.  The fill reads the
        // streams at those offsets, so an (untrusted) chunk whose streams are shorter than its maps claim is refused HERE, before anything
        // is read past a buffer or written into the image planes.
        e = hipGetLastError();
        uint32_t tot[5] = {};
        if (e == hipSuccess) e = hipMemcpyAsync(tot, totals, sizeof tot, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) break;
        bool fits = tileBase + tot[0] <= nTiles;
        for (int f = 0; f < 4; f++) fits = fits && (ib[f] + tot[1 + f] <= idxBytes[f]);
        if (!fits) { shortStream = true; break; }
        hipLaunchKernelGGL(yk_dl_tile_kernel<true>, dim3((unsigned)nb), dim3(1024), 0, c->stream, YK_DL_ARGS);
#undef YK_DL_ARGS
        hipLaunchKernelGGL(yk_dl_mark_kernel, dim3((unsigned)((nSlots + 255) / 256)), dim3(256), 0, c->stream, dMap, nSlots, g, w, h, reinterpret_cast<uint32_t*>(c->dTile4), (w + 15) >> 4);
        e = hipGetLastError();
        tileBase += tot[0];
        for (int f = 0; f < 4; f++) ib[f] += tot[1 + f];
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    freeAll();
    if (e != hipSuccess) return yk_fail(c, YK_ERR_HIP, "3-D LUT decode", e);
    if (shortStream) return yk_fail(c, YK_ERR_RANGE, "tile, colour or index stream shorter than the tile maps need");
    if (consumed) { consumed[0] = (size_t)tileBase * 2; consumed[1] = (size_t)tileBase * 6; for (int f = 0; f < 4; f++) consumed[2 + f] = (size_t)ib[f]; }
    return YK_OK;
}

}  // extern "C"
