// yk_encode2.hip — second-generation fused kernel (same results as the first-generation yk_encode_kernel of tests/csrc/yk_encode_v1.hip, different work decomposition).
//
//   a6       7x EncoderContext::FittingQuadSmooth            (encoder/EncoderContext.cpp:3710-4363)
//   a10-a13  DynamicTileEncode / GetMinMax_Y / GetTileDynamic_Y / DynamicTile::buildTable
//                                                             (encoder/EncoderContext.cpp:625-1212, 4365-4602; Plane.cpp:489)
//
// Work decomposition: one wave64 per workgroup; a unit is a 64x16 strip = four 16x16 macro-tiles = a quarter of a 64x64
// swizzle block; a LANE owns a whole 4x4 CELL (16 pixels, packed 0x00BBGGRR in 16 VGPRs) for the whole kernel
// (lane = macroTile*16 + cellY*4 + cellX).  Consequences:
//   * nothing waits on another wave: no workgroup barriers, ~8.5 KB LDS and 128 VGPRs per wave = 4 waves per SIMD;
//   * the gradient passes run in packed 16-bit arithmetic (two streams per VALU op) on corner values taken from a per-strip lattice
//     table (Round6 / Round6P precomputed once per lattice point); the per-tile set-up is amortised over 16 pixels per lane;
//   * a 4x4 tile is one lane, an 8x8 tile four lanes: tile-level reductions are ballots against a lane mask or two shuffles;
//   * every pass first tests ONE row of every cell; if that already rejects every viable tile of the wave (noise, mild noise)
//     the other three rows are never evaluated;
//   * range quantiser: index and minDiff of the nearest LUT entry of all six modes come from ONE gather per pixel out of a
//     per-device table indexed by (rangeDecode, v - BN) — every LUT the reference builds is BN + K[rangeDecode][mode];
//   * the mode-selection sums are summed in tree order and accepted only when a rigorous rounding margin separates the
//     modes; ambiguous tiles (rare) are re-summed in the reference's exact sequential order;
//   * the grid is XCD-aware (runs of 16 blocks per XCD, rotated per group) so halo lines are served by the neighbour's L2.
#include "yk_common.h"
#include "yk_curves.h"
#include "yk_device.h"
#include <mutex>

__constant__ float c_curve2[6][16] = YK_CURVE_TABLE;

#define LS YK_LSTRIDE
// timing ablations / the forced exact path (yk_set_ablation, include/yaik_hip_test.h) exist in the test build only
#ifdef YK_TEST_HOOKS
#define YK2_ABLATE(bit) ((P.ablate & (bit)) != 0)
#else
#define YK2_ABLATE(bit) false
#endif

__device__ __forceinline__ int y2_byte(uint32_t w, int ch) { return (w >> (8 * ch)) & 255; }
// |a - b| through the SAD unit (with a literal 0 addend the compiler would expand __usad into min/max/sub)
__device__ __forceinline__ uint32_t y2_absdiff(uint32_t a, uint32_t b) { uint32_t r; asm("v_sad_u32 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ int y2_round6(int v) { return (v & ~3) | (v >> 6); }                       // EncoderContext.cpp:3183
__device__ __forceinline__ int y2_round6p(int v) { v = min(v + 1, 255); return (v & ~3) | (v >> 6); } // EncoderContext.cpp:3202

// One gradient pass for the four macro-tiles of a wave.  Arithmetic identical to yk_grad_pass (tests/csrc/yk_encode_v1.hip): with
// 1/16-unit weights S' = (TL*lx+TR*rx)*wy + (BL*lx+BR*rx)*wb fits 16 bits and the six variants of EncoderContext.cpp:3929-3991
// are range tests on D = S' - 256*cur per corner set.  Here the arithmetic is PACKED: two 16-bit streams per VALU op.
//   * streams: t = 0..2: channel t of (raw corners | Round6 corners); t = 3: Round6P corners of (channel 0 | channel 1);
//     t = 4: Round6P corners of channel 2 (both halves).  The corner values come from a per-strip lattice table in LDS that is
//     already in this layout (s_lat[t][17x5 lattice points], built once per strip).
//   * S' is bilinear inside a tile, so a stream is (S, step) at one pixel plus three constants, all in the ring Z/2^16 (the true
//     S' lies in [0, 65280], so the ring value is exact): rows are walked in serpentine order with one packed subtract per pixel.
//   * D = S' - 256*cur is formed with signed saturation on operands biased by -32768 (pixel bytes ^ 0x80, S' ^ 0x8000); the
//     bounds of the tests are at most 256*rf + 255 = 16639 for the largest rejectFactor the C-ABI admits (64), well inside
//     int16, so a saturated D compares like the true D.  Only min D / max D per stream are tracked.
typedef short y2s2 __attribute__((ext_vector_type(2)));
typedef unsigned short y2u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ y2u2 y2_u2(uint32_t w) { return __builtin_bit_cast(y2u2, w); }
__device__ __forceinline__ y2s2 y2_s2(y2u2 v) { return __builtin_bit_cast(y2s2, v); }
__device__ __forceinline__ y2u2 y2_splat(int v) { const unsigned short t = (unsigned short)v; return (y2u2){ t, t }; }
__device__ __forceinline__ y2s2 y2_splats(int v) { const short t = (short)v; return (y2s2){ t, t }; }
#define YK2_LATN 85                                                      // 17 x 5 lattice points (every 4th pixel incl. the halo) per strip
// -DYK2_STATS (tools/path_stats.py only, never shipped): how often every path of the kernel is taken, summed over the launch
#ifdef YK2_STATS
__device__ unsigned long long g_y2_stats[128];
#define YK2_STAT(i, n) do { if (lane == 0) atomicAdd(&g_y2_stats[(i)], (unsigned long long)(n)); } while (0)
extern "C" int yk_debug_path_stats(unsigned long long* out, int clear) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_y2_stats), sizeof(g_y2_stats));
    if (e == hipSuccess && clear) { static unsigned long long z[128]; e = hipMemcpyToSymbol(HIP_SYMBOL(g_y2_stats), z, sizeof(z)); }
    return (int)e;
}
#else
#define YK2_STAT(i, n) do { } while (0)
#endif

// Tile-level decisions as wave-uniform 64-bit lane masks on the scalar unit.  Lane order inside a macro-tile is MORTON:
// lane = macroTile*16 + (cellY>>1)*8 + (cellX>>1)*4 + (cellY&1)*2 + (cellX&1), so that an 8x8 tile (2x2 cells) is one quad of lanes and its
// reductions are quad_perm DPP operands of the ALU instruction itself.  A tile of NX x NY cells is represented by the bit of its origin cell's lane.
__device__ __forceinline__ constexpr int y2_cell_x(int m) { return (m & 1) | ((m >> 1) & 2); }
__device__ __forceinline__ constexpr int y2_cell_y(int m) { return ((m >> 1) & 1) | ((m >> 2) & 2); }
template <int NX, int NY> __device__ __forceinline__ constexpr unsigned long long y2_origin() {      // lanes that are tile origins
    unsigned long long m = 0;
    for (int i = 0; i < 16; i++) if ((y2_cell_x(i) % NX) == 0 && (y2_cell_y(i) % NY) == 0) m |= 1ULL << i;
    return m * 0x0001000100010001ULL;
}
template <int NX, int NY> __device__ __forceinline__ unsigned long long y2_fold(unsigned long long m) {   // origin bit <- OR over the tile's lanes
    if (NX >= 2) m |= m >> 1;
    if (NY >= 2) m |= m >> 2;
    if (NX == 4) m |= m >> 4;
    if (NY == 4) m |= m >> 8;
    return m;                                                            // only the origin bits are meaningful
}
template <int NX, int NY> __device__ __forceinline__ unsigned long long y2_spread(unsigned long long m) { // origin bits -> all lanes of the tile
    if (NX >= 2) m |= m << 1;
    if (NY >= 2) m |= m << 2;
    if (NX == 4) m |= m << 4;
    if (NY == 4) m |= m << 8;
    return m;
}

template <int SX, int SY, bool SMOOTH = false>
__device__ __forceinline__ void y2_grad_pass(const uint32_t* s_lat, const int lat, const int cx, const int cy,
                                             const uint32_t (&pwb)[16], unsigned long long& cov, const unsigned long long deadLanes, const bool stripInside,
                                             const int gxCell, const int gyCell, const int w, const int h, const int rf,
                                             uint32_t* s_bm, const int bxCell, const int byCell, const uint32_t* s_pix, uint8_t* s_list, uint32_t* s_tf, const int lane) {
    constexpr int TX = 1 << SX, TY = 1 << SY, NX = TX / 4, NY = TY / 4;
    constexpr unsigned long long ORG = y2_origin<NX, NY>();
    const int dcx = cx & (NX - 1), dcy = cy & (NY - 1);                  // this cell's offset inside its tile, in cells
    // viable tiles (origin bits): top-left pixel uncovered (:3871-3875), no cell killed by the curvature test, whole tile inside the image
    unsigned long long viable = ORG & ~cov & ~y2_fold<NX, NY>(deadLanes);
    if (!stripInside) {
        const int tgx = gxCell - dcx * 4, tgy = gyCell - dcy * 4;        // tile origin, stripe-local pixels
        viable &= __ballot((tgx + TX <= w) && (tgy + TY <= h));
    }
    if (viable == 0ULL) return;
    constexpr int kPassId = (SX == 4 && SY == 4) ? 0 : (SX == 4 && SY == 3) ? 1 : (SX == 3 && SY == 4) ? 2 : (SX == 3 && SY == 3) ? 3 : (SX == 3 && SY == 2) ? 4 : (SX == 2 && SY == 3) ? 5 : 6;
    YK2_STAT(kPassId * 10 + 0, 1); YK2_STAT(kPassId * 10 + 8, __popcll(y2_spread<NX, NY>(viable)));
    const int loO = -256 * rf, hiO = 256 * rf + 255;
    const y2s2 loO2 = y2_splats(loO), hiO2 = y2_splats(hiO), loR2 = y2_splats(loO - 127), hiR2 = y2_splats(hiO - 127);


#ifndef YK2_NO_COMPACT
    // ---- few viable cells (a strip whose larger tiles failed along an edge, the usual case next to contours): a pass over 64 lanes
    // would keep at most a quarter of them busy for four rows.  Instead lane (j, r) = (j-th viable cell, row r of it) evaluates ONE row
    // of a viable cell; the pixels come from the staged strip in LDS.  A lane's failures are six bits (one per variant) OR-ed into a
    // word of its tile in LDS; a tile is lost when all six are set.  Two stages like the full pass: streams {0, 3} first (a tile whose
    // far corners sit on other content fails there), the other three only for waves that still hold a tile afterwards.
    const unsigned long long cells = y2_spread<NX, NY>(viable);
    const int nCells = __popcll(cells);
    if (nCells <= 16) {
        YK2_STAT(kPassId * 10 + 1, 1); YK2_STAT(kPassId * 10 + 7, nCells);
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(cells >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cells, 0u));
        const bool mine = __builtin_amdgcn_inverse_ballot_w64(cells);
        if (mine) s_list[rank] = (uint8_t)lane;
        s_tf[lane] = 0u; s_tf[64 + lane] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int j = lane & 15, r = lane >> 4;
        const bool act = j < nCells;
        const int c = (int)s_list[act ? j : 0];
        const int q2 = c >> 4, cy2 = y2_cell_y(c & 15), cx2 = y2_cell_x(c & 15);
        const int dcx2 = cx2 & (NX - 1), dcy2 = cy2 & (NY - 1);
        constexpr int kInTile = (NX > 1 ? 1 : 0) | (NY > 1 ? 2 : 0) | (NX > 2 ? 4 : 0) | (NY > 2 ? 8 : 0);   // Morton bits of a cell inside its tile
        const int orig = c & ~kInTile;                                       // lane of the tile's origin cell
        const int lo2 = cy2 * 17 + q2 * 4 + cx2 - dcy2 * 17 - dcx2;          // lattice index of the tile origin
        const y2u2 wyr = y2_splat(16 - ((dcy2 * 4 + r) << (4 - SY)));       // weight of row r of the cell
        const y2u2 lxc = y2_splat(16 - ((dcx2 * 4) << (4 - SX)));           // weight of the cell's first pixel column
        const uint4 pr = *reinterpret_cast<const uint4*>(&s_pix[(cy2 * 4 + r) * LS + q2 * 16 + cx2 * 4]);
        const uint32_t px[4] = { pr.x ^ 0x00808080u, pr.y ^ 0x00808080u, pr.z ^ 0x00808080u, pr.w ^ 0x00808080u };   // bias of the packed arithmetic
        y2s2 mnc[5], mxc[5];
        auto rowC = [&](const unsigned mask) {                               // the row's four pixels for the streams in `mask`
            y2u2 Sc[5], stc[5];
#pragma unroll
            for (int t = 0; t < 5; t++) {
                if (!((mask >> t) & 1u)) continue;
                const uint32_t* lt = s_lat + t * YK2_LATN + lo2;
                const y2u2 TL = y2_u2(lt[0]), TR = y2_u2(lt[NX]), BL = y2_u2(lt[NY * 17]), BR = y2_u2(lt[NY * 17 + NX]);
                const y2u2 e16 = (BL - BR) << 4, g = TR - BR, f = TL - BL - g;
                const y2u2 c2 = (g << 4) + f * lxc;
                Sc[t] = ((BR << 8) + e16 * lxc + c2 * wyr) ^ y2_splat(0x8000);   // S'(x0, row r), biased
                stc[t] = (e16 + f * wyr) << (4 - SX);                            // S'(x, r) - S'(x+1, r)
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t p = px[i];
                const y2s2 c01 = y2_s2(y2_u2(__builtin_amdgcn_perm(0u, p, 0x010C000Cu))), c22 = y2_s2(y2_u2(__builtin_amdgcn_perm(0u, p, 0x020C020Cu)));
                const y2s2 cc[4] = { __builtin_shufflevector(c01, c01, 0, 0), __builtin_shufflevector(c01, c01, 1, 1), c22, c01 };
#pragma unroll
                for (int t = 0; t < 5; t++) {
                    if (!((mask >> t) & 1u)) continue;
                    const y2s2 D = __builtin_elementwise_sub_sat(y2_s2(Sc[t]), cc[t == 4 ? 2 : t]);
                    if (i == 0) { mnc[t] = D; mxc[t] = D; }
                    else { mnc[t] = __builtin_elementwise_min(mnc[t], D); mxc[t] = __builtin_elementwise_max(mxc[t], D); }
                    if (i < 3) Sc[t] = Sc[t] - stc[t];
                }
            }
        };
        // the lane's failing variants as bits 31, 30 (raw / Round6 corners without the rounding term), 15, 14 (with it), 29, 13 (Round6P corners)
        constexpr uint32_t kAllFail = 0xE000E000u;
        auto failBits = [&](const y2s2 mnA, const y2s2 mxA, const y2s2 mnP, const y2s2 mxP) -> uint32_t {
            const y2s2 zero = y2_splats(0);
            // a half is negative when its variant has a pixel outside the window (x = raw corners, y = Round6 corners)
            const y2s2 oA = __builtin_elementwise_sub_sat(zero, __builtin_elementwise_max(__builtin_elementwise_sub_sat(loO2, mnA), __builtin_elementwise_sub_sat(mxA, hiO2)));
            const y2s2 rA = __builtin_elementwise_sub_sat(zero, __builtin_elementwise_max(__builtin_elementwise_sub_sat(loR2, mnA), __builtin_elementwise_sub_sat(mxA, hiR2)));
            const y2s2 oP = __builtin_elementwise_sub_sat(zero, __builtin_elementwise_max(__builtin_elementwise_sub_sat(loO2, mnP), __builtin_elementwise_sub_sat(mxP, hiO2)));
            const y2s2 rP = __builtin_elementwise_sub_sat(zero, __builtin_elementwise_max(__builtin_elementwise_sub_sat(loR2, mnP), __builtin_elementwise_sub_sat(mxP, hiR2)));
            const uint32_t a = __builtin_bit_cast(uint32_t, oA), b = __builtin_bit_cast(uint32_t, rA), cO = __builtin_bit_cast(uint32_t, oP), cR = __builtin_bit_cast(uint32_t, rP);
            // raw-O -> bit 15, Round6-O -> bit 31, raw-R -> 14, Round6-R -> 30, Round6P-O (either half) -> 29, Round6P-R -> 13
            uint32_t m = (a & 0x80008000u) | ((b >> 1) & 0x40004000u);
            m |= (((cO << 16) | cO) >> 2) & 0x20000000u;
            m |= (((cR >> 16) | cR) >> 2) & 0x00002000u;
            return m;
        };
        rowC(0x09u);
        if (act) atomicOr(&s_tf[orig], failBits(mnc[0], mxc[0], mnc[3], mxc[3]));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const unsigned long long alive = viable & __ballot(s_tf[lane] != kAllFail);
        if (alive == 0ULL) { YK2_STAT(100 + kPassId, 1); return; }
        rowC(0x16u);
        {
            const y2s2 mnA = __builtin_elementwise_min(__builtin_elementwise_min(mnc[0], mnc[1]), mnc[2]), mxA = __builtin_elementwise_max(__builtin_elementwise_max(mxc[0], mxc[1]), mxc[2]);
            const y2s2 mnP = __builtin_elementwise_min(mnc[3], mnc[4]), mxP = __builtin_elementwise_max(mxc[3], mxc[4]);
            if (act) atomicOr(&s_tf[64 + orig], failBits(mnA, mxA, mnP, mxP));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const unsigned long long acceptC = alive & __ballot(s_tf[64 + lane] != kAllFail);   // some variant never failed (:3998)
        if (acceptC != 0ULL) {
            YK2_STAT(110 + kPassId, 1);
            cov |= y2_spread<NX, NY>(acceptC);
            if (__builtin_amdgcn_inverse_ballot_w64(acceptC)) {
                const int tbx = bxCell >> SX, tby = byCell >> SY;
                int bit;
                if (SX == 4 && SY == 4) bit = 0 + tby * 4 + tbx;
                else if (SX == 4 && SY == 3) bit = 32 + tby * 4 + tbx;
                else if (SX == 3 && SY == 4) bit = 64 + tby * 8 + tbx;
                else if (SX == 3 && SY == 3) bit = 96 + tby * 8 + tbx;
                else if (SX == 3 && SY == 2) bit = 160 + (tby >> 3) * 64 + (tby & 7) * 8 + tbx;
                else if (SX == 2 && SY == 3) bit = 288 + (tbx >> 3) * 64 + tby * 8 + (tbx & 7);
                else bit = 416 + ((tby >> 3) * 2 + (tbx >> 3)) * 64 + (tby & 7) * 8 + (tbx & 7);
                atomicOr(&s_bm[bit >> 5], 1u << (bit & 31));
            }
        }
        return;
    }
#endif

    // per stream: S (at the current pixel, biased), step along x, and the constants of the walk
    const int lo = lat - dcy * 17 - dcx;                                 // lattice index of the tile origin
    const y2u2 wy2 = y2_splat(16 - ((dcy * 4) << (4 - SY)));             // weight of the cell's first pixel row
    const y2u2 lx2 = y2_splat(16 - ((dcx * 4) << (4 - SX)));             // weight of the cell's first pixel column
    y2u2 S[5], st[5], dS0[5], dS3[5], dst[5];
    auto setup = [&](const int t) {
        const uint32_t* lt = s_lat + t * YK2_LATN + lo;
        const y2u2 TL = y2_u2(lt[0]), TR = y2_u2(lt[NX]), BL = y2_u2(lt[NY * 17]), BR = y2_u2(lt[NY * 17 + NX]);
        // L(r) = 16 BL + (TL-BL) wy, R(r) = 16 BR + (TR-BR) wy, dL = L - R, S'(x0) = 16 R + dL lx0, step = dL * 16/TX, wy(r) = wy0 - r * 16/TY
        const y2u2 e16 = (BL - BR) << 4, g = TR - BR, f = TL - BL - g;
        const y2u2 c2 = (g << 4) + f * lx2;
        S[t] = ((BR << 8) + e16 * lx2 + c2 * wy2) ^ y2_splat(0x8000);
        dS0[t] = c2 << (4 - SY);                                         // S'(x0, r) - S'(x0, r+1)
        st[t] = (e16 + f * wy2) << (4 - SX);                             // S'(x, r) - S'(x+1, r)
        dst[t] = f << (8 - SX - SY);                                     // step(r) - step(r+1)
        dS3[t] = dS0[t] - dst[t] * (unsigned short)3;                    // S'(x0+3, r) - S'(x0+3, r+1)
    };
    y2s2 mn[5], mx[5];
    // one row of the cell for the streams in `mask`; `first`: these streams' running extremes start with this row's first pixel
    auto pixelRow = [&](const int r, const unsigned mask, const bool first) {
#pragma unroll
        for (int ii = 0; ii < 4; ii++) {
            const int i = (r & 1) ? 3 - ii : ii;                         // serpentine: odd rows right to left
            const uint32_t p = pwb[r * 4 + i];
            // 256 * (cur - 128) of channel c in both halves; stream 3: channel 0 | channel 1
            // two byte permutations per pixel; the broadcasts of one half are operand selects (op_sel) of the packed subtract
            const y2s2 c01 = y2_s2(y2_u2(__builtin_amdgcn_perm(0u, p, 0x010C000Cu))), c22 = y2_s2(y2_u2(__builtin_amdgcn_perm(0u, p, 0x020C020Cu)));
            const y2s2 cc[4] = { __builtin_shufflevector(c01, c01, 0, 0), __builtin_shufflevector(c01, c01, 1, 1), c22, c01 };
#pragma unroll
            for (int t = 0; t < 5; t++) {
                if (!((mask >> t) & 1u)) continue;
                const y2s2 D = __builtin_elementwise_sub_sat(y2_s2(S[t]), cc[t == 4 ? 2 : t]);
                if (first && ii == 0) { mn[t] = D; mx[t] = D; }
                else { mn[t] = __builtin_elementwise_min(mn[t], D); mx[t] = __builtin_elementwise_max(mx[t], D); }
                if (ii < 3) S[t] = (r & 1) ? S[t] + st[t] : S[t] - st[t];
            }
        }
    };
    auto nextRow = [&](const int r, const unsigned mask) {               // from the end of row r to the start of row r + 1, streams in `mask`
#pragma unroll
        for (int t = 0; t < 5; t++) if ((mask >> t) & 1u) { S[t] -= (r & 1) ? dS0[t] : dS3[t]; st[t] -= dst[t]; }
    };
    // Lanes failing a variant -> tiles (origin bits) with a failing lane.  The six variants of :3929-3991 are range tests on the running
    // extremes: A = the (raw | Round6) streams merged over their channels, halves x / y = raw / Round6 corners; P = the Round6P streams
    // (either half); O / R = the window without / with the rounding term.
    auto failA4 = [&](const y2s2 mnA, const y2s2 mxA) -> unsigned long long {      // tiles in which raw-O, raw-R, Round6-O and Round6-R all have a failing lane
        const y2s2 oA = __builtin_elementwise_max(__builtin_elementwise_sub_sat(loO2, mnA), __builtin_elementwise_sub_sat(mxA, hiO2));
        const y2s2 rA = __builtin_elementwise_max(__builtin_elementwise_sub_sat(loR2, mnA), __builtin_elementwise_sub_sat(mxA, hiR2));
        unsigned long long f = y2_fold<NX, NY>(__ballot(oA.x > 0));
        f &= y2_fold<NX, NY>(__ballot(oA.y > 0));
        f &= y2_fold<NX, NY>(__ballot(rA.x > 0));
        f &= y2_fold<NX, NY>(__ballot(rA.y > 0));
        return f;
    };
    auto failP2 = [&](const y2s2 mnP, const y2s2 mxP) -> unsigned long long {      // tiles in which Round6P-O and Round6P-R both have a failing lane
        const y2s2 oP = __builtin_elementwise_max(__builtin_elementwise_sub_sat(loO2, mnP), __builtin_elementwise_sub_sat(mxP, hiO2));
        const y2s2 rP = __builtin_elementwise_max(__builtin_elementwise_sub_sat(loR2, mnP), __builtin_elementwise_sub_sat(mxP, hiR2));
        return y2_fold<NX, NY>(__ballot((oP.x > 0) | (oP.y > 0))) & y2_fold<NX, NY>(__ballot((rP.x > 0) | (rP.y > 0)));
    };
    auto mnAll = [&]() { return __builtin_elementwise_min(__builtin_elementwise_min(mn[0], mn[1]), mn[2]); };   // (raw | Round6) over the channels
    auto mxAll = [&]() { return __builtin_elementwise_max(__builtin_elementwise_max(mx[0], mx[1]), mx[2]); };

    // Order of evaluation (exact either way: a tile is dropped only when each of its six variants has a failing pixel, and accepted
    // only when some variant has none over the whole tile):
    //  * row 0 of every cell first; tiles of four or more cells are screened with two of the five streams (channel 0 of the raw and
    //    Round6 corners, channels 0 and 1 of the Round6P corners): on noisy content that already fails every variant of every tile
    //    of the wave, and the other streams are never set up;
    //  * the (raw | Round6) streams run ahead over the four rows; the Round6P streams only finish for waves that hold a tile none of
    //    the four raw / Round6 variants accepts (on clean gradients the raw corners pass and two streams in five are never walked).
    constexpr bool kScreen = (NX * NY >= 4);
    constexpr unsigned kA = 0x07u, kP = 0x18u;
    unsigned long long accept;
    if (SMOOTH) {
        // A strip without a single lane killed by the curvature test (smooth content): the screens below would only cost it instructions and
        // two more dependent round trips through LDS and the scalar unit.  The raw / Round6 streams walk the whole tile at once; the Round6P
        // streams only for tiles those four variants fail.
        setup(0); setup(1); setup(2);
        pixelRow(0, kA, true);
        nextRow(0, kA); pixelRow(1, kA, false);
        nextRow(1, kA); pixelRow(2, kA, false);
        nextRow(2, kA); pixelRow(3, kA, false);
        YK2_STAT(kPassId * 10 + 4, 1);
        accept = viable & ~failA4(mnAll(), mxAll());
#ifdef YK2_STATS
        {   // which variants accept the tiles of smooth strips (never shipped)
            y2u2 S_[5], st_[5], a_[5], b_[5], c_[5]; y2s2 mn_[5], mx_[5];
            for (int t = 0; t < 5; t++) { S_[t] = S[t]; st_[t] = st[t]; a_[t] = dS0[t]; b_[t] = dS3[t]; c_[t] = dst[t]; mn_[t] = mn[t]; mx_[t] = mx[t]; }
            setup(3); setup(4);
            pixelRow(0, kP, true); nextRow(0, kP); pixelRow(1, kP, false); nextRow(1, kP); pixelRow(2, kP, false); nextRow(2, kP); pixelRow(3, kP, false);
            const unsigned long long accP = viable & ~failP2(__builtin_elementwise_min(mn[3], mn[4]), __builtin_elementwise_max(mx[3], mx[4]));
            YK2_STAT(97, __popcll(accept)); YK2_STAT(98, __popcll(accP)); YK2_STAT(99, __popcll(accept | accP)); YK2_STAT(96 + 0 * kPassId, 0);
            YK2_STAT(107, __popcll(viable));
            for (int t = 0; t < 5; t++) { S[t] = S_[t]; st[t] = st_[t]; dS0[t] = a_[t]; dS3[t] = b_[t]; dst[t] = c_[t]; mn[t] = mn_[t]; mx[t] = mx_[t]; }
        }
#endif
        const unsigned long long rest = viable & ~accept;
        if (rest != 0ULL) {
            YK2_STAT(kPassId * 10 + 5, 1); YK2_STAT(90 + kPassId, __popcll(y2_spread<NX, NY>(rest)));
            // a tile next to a contour usually fails these two variants in the first row of its cells already (its corners are off)
            setup(3);
            pixelRow(0, 0x08u, true);
            const unsigned long long left = rest & ~failP2(mn[3], mx[3]);
            if (left != 0ULL) {
                setup(4);
                pixelRow(0, 0x10u, true);
                nextRow(0, kP); pixelRow(1, kP, false);
                nextRow(1, kP); pixelRow(2, kP, false);
                nextRow(2, kP); pixelRow(3, kP, false);
                accept |= left & ~failP2(__builtin_elementwise_min(mn[3], mn[4]), __builtin_elementwise_max(mx[3], mx[4]));
            }
        }
    } else {
    int pRows;                                                           // rows the Round6P streams have walked so far
    if (kScreen) {
        setup(0); setup(3);
        pixelRow(0, 0x09u, true);
        viable &= ~(failA4(mn[0], mx[0]) & failP2(mn[3], mx[3]));
        if (viable == 0ULL) { YK2_STAT(kPassId * 10 + 2, 1); return; }
        setup(1); setup(2);
        pixelRow(0, 0x06u, true);
        viable &= ~(failA4(mnAll(), mxAll()) & failP2(mn[3], mx[3]));      // after one row: lost tiles cannot be accepted by later rows
        if (viable == 0ULL) { YK2_STAT(kPassId * 10 + 3, 1); return; }
        nextRow(0, kA);
        pixelRow(1, kA, false);
        pRows = 0;                                                       // stream 3 has walked row 0, stream 4 nothing yet
    } else {
#pragma unroll
        for (int t = 0; t < 5; t++) setup(t);
        pixelRow(0, 0x1Fu, true);
        viable &= ~(failA4(mnAll(), mxAll()) & failP2(__builtin_elementwise_min(mn[3], mn[4]), __builtin_elementwise_max(mx[3], mx[4])));
        if (viable == 0ULL) { YK2_STAT(kPassId * 10 + 3, 1); return; }
        nextRow(0, 0x1Fu);
        pixelRow(1, 0x1Fu, false);
        if (NX * NY == 1) {                                              // 4x4 tiles: one lane per tile, a second look after half of the tile
            viable &= ~(failA4(mnAll(), mxAll()) & failP2(__builtin_elementwise_min(mn[3], mn[4]), __builtin_elementwise_max(mx[3], mx[4])));
            if (viable == 0ULL) { YK2_STAT(kPassId * 10 + 9, 1); return; }
        }
        pRows = 2;
    }
    YK2_STAT(kPassId * 10 + 4, 1);
    nextRow(1, kA);
    pixelRow(2, kA, false);
    nextRow(2, kA);
    pixelRow(3, kA, false);
    accept = viable & ~failA4(mnAll(), mxAll());                         // some raw / Round6 variant never failed (:3998)
    const unsigned long long rest = viable & ~accept;
    if (rest != 0ULL) {                                                  // wave-uniform: the Round6P variants decide the remaining tiles
        YK2_STAT(kPassId * 10 + 5, 1); YK2_STAT(90 + kPassId, __popcll(y2_spread<NX, NY>(rest)));
        if (kScreen) {
            setup(4);
            pixelRow(0, 0x10u, true);
            nextRow(0, kP);
            pixelRow(1, kP, false);
        }
        nextRow(1, kP);
        pixelRow(2, kP, false);
        nextRow(2, kP);
        pixelRow(3, kP, false);
        accept |= rest & ~failP2(__builtin_elementwise_min(mn[3], mn[4]), __builtin_elementwise_max(mx[3], mx[4]));
    }
    (void)pRows;
    }
    if (accept == 0ULL) return;
    YK2_STAT(kPassId * 10 + 6, 1);
    cov |= y2_spread<NX, NY>(accept);                                    // paint coverage (:4029-4037): bit = lane = cell
    if (__builtin_amdgcn_inverse_ballot_w64(accept)) {                   // the origin cell's lane sets the bitmap bit (:4026)
        const int tbx = bxCell >> SX, tby = byCell >> SY;                // tile coordinates inside the 64x64 block
        int bit;
        if (SX == 4 && SY == 4) bit = 0 + tby * 4 + tbx;
        else if (SX == 4 && SY == 3) bit = 32 + tby * 4 + tbx;
        else if (SX == 3 && SY == 4) bit = 64 + tby * 8 + tbx;
        else if (SX == 3 && SY == 3) bit = 96 + tby * 8 + tbx;
        else if (SX == 3 && SY == 2) bit = 160 + (tby >> 3) * 64 + (tby & 7) * 8 + tbx;
        else if (SX == 2 && SY == 3) bit = 288 + (tbx >> 3) * 64 + tby * 8 + (tbx & 7);
        else bit = 416 + ((tby >> 3) * 2 + (tbx >> 3)) * 64 + (tby & 7) * 8 + (tbx & 7);
        atomicOr(&s_bm[bit >> 5], 1u << (bit & 31));
    }
}

// ---- quantiser table: for every rangeDecode R (32..255) and every offset o = v - BN (-2..255) the index and minDiff of the first
// nearest entry of the six LUTs BN + K[R][mode][i], K = trunc(curve * R) (DynamicTile::buildTable, EncoderContext.cpp:662-696, and
// the `<` scan of GetTileDynamic_Y, :873-881).  Row = 16 bytes: minDiff of modes 0..5 as six f16 (exact integers; v_fma_mix_f32 takes
// them as they are, a byte would need a conversion instruction per mode and pixel) | six index nibbles | 0.
#define YK2_QROWS 258
#define YK2_QR0 32
#define YK2_QNR 224
#define YK2_QBYTES ((size_t)YK2_QNR * YK2_QROWS * 16)
#define YK2_RCPBYTES ((size_t)256 * sizeof(float))
// ---- tile-definition table: DynamicTile::buildTable's integer part (:625-661) for every (min_, d8) a tile can have, min_ = min(min, 224) in
// 0..224, d8 = max(max - min_, 32) in 32..255: [first row of the tile's slab in the quantiser table + 256 - BN + 2 (16 bits) | EncodeTileType's
// range and base fields (15 bits)].  One 4-byte gather per tile-plane instead of ~40 instructions of divisions by constants per lane.
#define YK2_DEFN1 224
#define YK2_DEFBYTES ((size_t)225 * YK2_DEFN1 * 4)
#define YK2_TABBYTES (YK2_QBYTES + YK2_RCPBYTES + YK2_DEFBYTES)
typedef _Float16 y2h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t y2_f16_bits(int n) { const _Float16 h = (_Float16)(float)n; return (uint32_t)__builtin_bit_cast(unsigned short, h); }
__global__ void yk_qtab_kernel(uint4* tab) {
    __shared__ int K[6][16];
    const int R = YK2_QR0 + (int)blockIdx.x, t = threadIdx.x;
    if (t < 96) { const int m = t >> 4, i = t & 15; K[m][i] = __float2int_rz(__fmul_rn(c_curve2[m][i], (float)R)); }
    __syncthreads();
    if (t < YK2_QROWS) {
        const int o = t - 2;
        uint32_t md[6], idx = 0;
        for (int m = 0; m < 6; m++) {
            const int cnt = m < 3 ? 16 : 8;
            int best = 1 << 30, bi = 0;
            for (int n = 0; n < cnt; n++) { const int d = abs(K[m][n] - o); if (d < best) { best = d; bi = n; } }
            best = min(best, 255);                                       // only offsets no tile can reach exceed a byte
            md[m] = y2_f16_bits(best);
            idx |= (uint32_t)bi << (4 * m);
        }
        tab[(size_t)blockIdx.x * YK2_QROWS + t] = make_uint4(md[0] | (md[1] << 16), md[2] | (md[3] << 16), md[4] | (md[5] << 16), idx);
    }
}
// the integer part of DynamicTile::buildTable for a tile whose valid pixels span [mn, mx] (EncoderContext.cpp:625-661)
struct Y2TileDef { int base, BN, dist, R; };
__device__ __forceinline__ Y2TileDef y2_tile_def(const int min_, const int d8) {
    Y2TileDef d;
    d.base = (min_ * 63 + 112) / 224;
    d.BN = (d.base * 224) / 63;
    const int scale = 223 - d.BN;
    const int dnum = (d8 - 32) * 127 + (scale - 1);
    d.dist = (scale < 0) ? -dnum : dnum / scale;
    d.R = (d.dist * scale) / 127 + 32;
    return d;
}
__global__ void yk_deftab_kernel(uint32_t* tab) {
    const int min_ = blockIdx.x, d8 = 32 + (int)threadIdx.x;
    const Y2TileDef d = y2_tile_def(min_, d8);
    const int R = min(max(d.R, YK2_QR0), YK2_QR0 + YK2_QNR - 1);         // pairs no tile can have (checked by yk_selftest 3) stay inside the table
    const uint32_t rowBase = (uint32_t)((R - YK2_QR0) * YK2_QROWS + 2 - d.BN + 256);
    tab[min_ * YK2_DEFN1 + (int)threadIdx.x] = rowBase | (((((uint32_t)d.dist & 255u) << 7) | ((uint32_t)d.base & 255u)) << 16);
}

// yk_selftest 3: for every (min, max) of a tile, the LUTs built the reference's way (float add of BN) equal BN + K[rangeDecode],
// every value of the tile finds, in the table, the entry a scan of those LUTs finds, and the tile-definition table holds the tile's slab and fields.
__global__ void yk_selftest_qtab_kernel(const uint4* tab, const uint32_t* deftab, int* mismatches) {
    const int mn = blockIdx.x, mx = threadIdx.x;
    if (mx < mn) return;
    const int min_ = min(mn, 224);
    int diff = mx - min_; if (diff < 16) diff = 16;
    const int d8 = max(diff, 32);
    const Y2TileDef d = y2_tile_def(min_, d8);
    const int BN = d.BN, R = d.R;
    int bad = 0;
    if (R < YK2_QR0 || R >= YK2_QR0 + YK2_QNR || mn - BN < -2) { atomicAdd(mismatches, 1); return; }
    {
        const uint32_t e = deftab[min_ * YK2_DEFN1 + (d8 - 32)];
        if ((e & 0xFFFFu) != (uint32_t)((R - YK2_QR0) * YK2_QROWS + 2 - BN + 256)) bad++;
        if ((e >> 16) != ((((uint32_t)d.dist & 255u) << 7) | ((uint32_t)d.base & 255u))) bad++;
    }
    for (int m = 0; m < 6; m++) {
        const int cnt = m < 3 ? 16 : 8;
        int L[16];
        for (int i = 0; i < cnt; i++) {
            L[i] = __float2int_rz(__fadd_rn((float)BN, __fmul_rn(c_curve2[m][i], (float)R)));
            if (L[i] != BN + __float2int_rz(__fmul_rn(c_curve2[m][i], (float)R))) bad++;
        }
        for (int v = mn; v <= mx; v++) {
            int best = 1 << 30, bi = 0;
            for (int n = 0; n < cnt; n++) { const int dd = abs(L[n] - v); if (dd < best) { best = dd; bi = n; } }
            const uint4 row = tab[(size_t)(R - YK2_QR0) * YK2_QROWS + (v - BN + 2)];
            const uint32_t w = m < 2 ? row.x : (m < 4 ? row.y : row.z);
            const uint32_t hb = (m & 1) ? (w >> 16) : (w & 0xFFFFu);
            const int ix = (row.w >> (4 * m)) & 15;
            if (hb != y2_f16_bits(best) || ix != bi) bad++;
        }
    }
    if (bad) atomicAdd(mismatches, bad);
}

#define YK2_RUN 16
// `old` with lane k replaced by the wave-uniform value `v` (v_writelane_b32 takes one scalar register: the lane is an inline constant; callers
// pass constants or unrolled loop counters, so the switch folds away)
template <int K> __device__ __forceinline__ uint32_t y2_writelane_k(uint32_t v, uint32_t old) { asm("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(v), "n"(K)); return old; }
__device__ __forceinline__ uint32_t y2_writelane(uint32_t v, int k, uint32_t old) {
    switch (k) {
        case 0: return y2_writelane_k<0>(v, old); case 1: return y2_writelane_k<1>(v, old); case 2: return y2_writelane_k<2>(v, old);
        case 3: return y2_writelane_k<3>(v, old); case 4: return y2_writelane_k<4>(v, old); default: return y2_writelane_k<5>(v, old);
    }
}
__device__ __forceinline__ uint32_t y2_sinkable(uint32_t v) { return v; }
__device__ __forceinline__ uint32_t y2_sinkable(uint2 v) { return v.x ^ v.y; }
// -DYK2_NOSTORE=<bits> (timing experiments only, never shipped): output stores of the fused kernel feed a checksum instead (the arithmetic stays):
// 1 = the nibble slots, 2 = per-tile counts and definitions, 4 = the atomics (16x16 map, scan sums), 8 = bitmaps, 16 = coverage
#ifdef YK2_NOSTORE
#define Y2_SINK(val) do { y2sink ^= (uint32_t)(val); } while (0)
#else
#define YK2_NOSTORE 0
#define Y2_SINK(val) do { } while (0)
#endif
#define Y2_STORE_IF(bit, lhs, val) do { if (YK2_NOSTORE & (bit)) Y2_SINK(y2_sinkable(val)); else lhs = (val); } while (0)
#define Y2_STORE(lhs, val) Y2_STORE_IF(2, lhs, val)
#define Y2_ATOMIC(call, val) do { if (YK2_NOSTORE & 4) Y2_SINK(val); else { call; } } while (0)
// -DYK2_TIMING (tools/wave_timeline.sh only, never shipped): every wave records the shader clock and the 100 MHz real-time counter at
// five points, its entry time and its hardware slot (HW_ID, XCC_ID) into a device array that yk_debug_wave_times copies out.
#ifdef YK2_TIMING
__device__ unsigned long long g_y2_times[65536 * 16];
#define YK2_PROBE(i) do { if ((i) == 0 && lane == 0 && unit < 65536) { g_y2_times[(size_t)unit * 16 + 10] = 0; g_y2_times[(size_t)unit * 16 + 11] = 0; g_y2_times[(size_t)unit * 16 + 12] = y2_t_entry; g_y2_times[(size_t)unit * 16 + 13] = ((unsigned long long)y2_xcc << 32) | y2_hwid; } if (lane == 0 && unit < 65536) { g_y2_times[(size_t)unit * 16 + 2 * (i)] = __builtin_amdgcn_s_memtime(); g_y2_times[(size_t)unit * 16 + 2 * (i) + 1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
extern "C" int yk_debug_wave_times(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_y2_times), sizeof(g_y2_times)); }
#else
#define YK2_PROBE(i) do { } while (0)
#endif
#define YK2_LUTW 84
#ifndef YK2_WIN
#define YK2_WIN 16
#endif
#ifndef YK2_PREF
#define YK2_PREF 8                                                       // rows of a plane requested before the previous plane is finished
#endif
#ifndef YK2_RVWIN
#define YK2_RVWIN 8
#endif

// values of the other lanes of the lane's quad (= its 8x8 tile) as DPP operands
#define Y2_QUAD_X 0xB1                                                   // quad_perm:[1,0,3,2]: the cell beside this one
#define Y2_QUAD_Y 0x4E                                                   // quad_perm:[2,3,0,1]: the cell above / below
template <int CTRL> __device__ __forceinline__ int y2_quad(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, true); }   // every lane is written: no initialising move
template <int CTRL> __device__ __forceinline__ float y2_quad(float v) { return __int_as_float(y2_quad<CTRL>(__float_as_int(v))); }
// t[m] += its values in the other three lanes of the quad, for modes FIRST..5: the DPP operand of the add itself.  One block of assembly: the
// vectoriser otherwise pairs the adds into v_pk_add_f32 (no DPP operand: two moves per value), and a DPP read needs two wait states behind the
// VALU write of its register, which the compiler does not track inside inline assembly: the block opens with s_nop 1 and a register is
// read again only after the adds of the other modes (at least two instructions).
template <int FIRST> __device__ __forceinline__ void y2_quad_sums(float (&t)[6]) {
#define Y2_QX(i) "v_add_f32_dpp %" #i ", %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define Y2_QY(i) "v_add_f32_dpp %" #i ", %" #i ", %" #i " quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
    if (FIRST == 0)
        asm volatile("s_nop 1\n\t" Y2_QX(0) Y2_QX(1) Y2_QX(2) Y2_QX(3) Y2_QX(4) Y2_QX(5) Y2_QY(0) Y2_QY(1) Y2_QY(2) Y2_QY(3) Y2_QY(4) Y2_QY(5)
                     : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]));
    else
        asm volatile("s_nop 1\n\t" Y2_QX(0) Y2_QX(1) Y2_QX(2) Y2_QY(0) Y2_QY(1) Y2_QY(2) : "+v"(t[3]), "+v"(t[4]), "+v"(t[5]));
#undef Y2_QX
#undef Y2_QY
}
// sum of a value over the 16 lanes of a DPP row (every lane ends with the same total: each step adds mirror-image partners)
__device__ __forceinline__ float y2_row16_sum(float x) {
    x = __fadd_rn(x, y2_quad<Y2_QUAD_X>(x));
    x = __fadd_rn(x, y2_quad<Y2_QUAD_Y>(x));
    x = __fadd_rn(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xF, 0xF, true)));   // row_half_mirror
    x = __fadd_rn(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xF, 0xF, true)));   // row_mirror
    return x;
}
// 4 * byte `ch` of w in one instruction (SDWA byte select feeding the shift)
__device__ __forceinline__ uint32_t y2_byte_x4(uint32_t w, int ch) {
    uint32_t r; const uint32_t two = 2u;
    if (ch == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(two), "v"(w));
    else if (ch == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(two), "v"(w));
    else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(two), "v"(w));
    return r;
}
// acc = fma((float)half(h), w, acc), half = low / high 16 bits of h as f16: v_fma_mix_f32 converts inside the operation
__device__ __forceinline__ void y2_fma_lo(float& acc, uint32_t h, float w) { asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(h), "v"(w)); }
__device__ __forceinline__ void y2_fma_hi(float& acc, uint32_t h, float w) { asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(h), "v"(w)); }
// R | G << 8 | B << 16 of three samples in 0..255 (two byte permutations)
__device__ __forceinline__ uint32_t y2_pack3(int r, int g, int b) {
    return __builtin_amdgcn_perm((uint32_t)b, __builtin_amdgcn_perm((uint32_t)g, (uint32_t)r, 0x0C0C0400u), 0x0C040100u);
}

// WANT_DST: the test-only reconstruction of the coded pixels into P.dst (yk_range_dst).  It is a separate instantiation because its extra
// live state costs the product kernel spilled VGPRs, and with them a scratch allocation per wave launch.
// MODE3: DynamicTileEncode's mode3BitOnly (modes 3..5 only).
template <bool WANT_DST, bool MODE3>
__global__ __launch_bounds__(64, 4) void yk_encode2_kernel(const YkEncodeParams P) {
    // One wave64 per workgroup: a work unit is a 64x16 strip (four macro-tiles) of a 64x64 swizzle block, so nothing in the
    // kernel waits on another wave.  The staged pixels are only read by the gradient passes (afterwards every lane holds its
    // 16 pixels in registers), so the test build's LUTs reuse the same LDS.
    // LUT layout per 8x8 tile (WANT_DST only): 4-bit mode m at [20m, 20m+16), 3-bit mode m at [60+8(m-3), +8); entries are LUT << 8.
    constexpr int kStart = MODE3 ? 3 : 0;
    constexpr int kPixWords = 17 * LS, kLutWords = WANT_DST ? 16 * YK2_LUTW : 0;
    __shared__ __attribute__((aligned(16))) uint32_t s_mem[kPixWords > kLutWords ? kPixWords : kLutWords];
    uint32_t* const s_pix = s_mem;
    uint32_t (*const s_lut)[YK2_LUTW] = reinterpret_cast<uint32_t (*)[YK2_LUTW]>(s_mem);
    __shared__ uint32_t s_bm[24];
    __shared__ uint8_t s_list[64];                                          // gradient passes with few viable cells: their lanes, compacted
    __shared__ __attribute__((aligned(16))) float s_curve[WANT_DST ? 6 : 1][16];
    __shared__ __attribute__((aligned(16))) float s_rcp[256];               // RN(1 / pixel value); [0] = 0 (skipped term, :884); correctly rounded: the exact path needs that
    // gradient phase: the corner lattice in stream layout (5 x 85 words); range phase: exact-order fallback (one tile-plane at a
    // time) and its mode sums
    __shared__ __attribute__((aligned(16))) uint32_t s_aux[432];
    uint32_t* const s_lat = s_aux;
    float (*const s_chain)[68] = reinterpret_cast<float (*)[68]>(s_aux);
    float* const s_err = reinterpret_cast<float*>(s_aux + 408);
    __shared__ uint32_t s_range[3 * 64];
    __shared__ __attribute__((aligned(16))) uint32_t s_sp[3 * 4 * 8 + 3 * 4 * 16];      // strips with at most four cells to code: six sums and sixteen index words per plane and cell

    uint32_t* const s_tile = s_aux + 416;                                    // exact-order fallback: the 64 values of the tile-plane being re-summed
    // range phase: the index words of a plane's sixteen rows per lane; the staged pixels are dead by then (the test build's LUTs live there)
    __shared__ uint32_t s_iwTest[WANT_DST ? 1024 : 1];
    uint32_t* const s_iw = WANT_DST ? s_iwTest : s_mem;
    static_assert(WANT_DST || kPixWords >= 1024, "index words need 4 KB");

    const int lane = threadIdx.x;
    // A new wave is the youngest of its SIMD: with equal priorities the arbiter lets the older waves' arithmetic go first and the 18
    // loads below leave only when those stall.  Until they are issued the wave runs at the highest priority (the ~60 instructions it
    // takes cost the others nothing measurable; the pixels arrive that much earlier: -2 % on the frame, -4 % on noise).
    __builtin_amdgcn_s_setprio(3);
#ifdef YK2_TIMING
    const unsigned long long y2_t_entry = __builtin_amdgcn_s_memrealtime();
    const unsigned y2_hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)), y2_xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
#endif
    // one 64x16 strip; `slot`, `xcd`: position in the XCD-aware unit order (below).  The kernel arguments are read through the kernel-argument
    // segment pointer where they are used (scalar loads from the constant cache) instead of being copied into SGPRs at the top: 16 spilled
    // SGPRs fewer.  (The lambda is what is left of the round-4 persistent-wave experiment, commit b2a2a39, profiles/r04/a*: DESIGN 5.3.)
    typedef const __attribute__((address_space(4))) YkEncodeParams* Y2ParamPtr;
    auto strip = [&](const int slot, const int xcd, const int lane, Y2ParamPtr const Pp) {
    const __attribute__((address_space(4))) YkEncodeParams& P = *Pp;
    const int nBf = P.xBB64 * P.yBB64, nB = nBf * P.nFrames;             // blocks per frame / of the whole batch
#if YK2_NOSTORE
    uint32_t y2sink = 0u;
#endif
    // XCD-aware unit order: consecutive workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2).  Units are
    // taken in row-major runs of YK2_RUN blocks (4 strips each); XCD k gets a rotating run of every group of 8 runs, so the
    // halo column of a block and the halo row of a strip are lines a neighbour streams through the same L2 at about the same
    // time instead of a second fabric fetch, while every XCD still samples the whole image (whole bands per XCD would leave
    // the cheapest band's XCD idle).
    const int grp = slot / (YK2_RUN * 4);
    const int unit = (grp * 8 + ((xcd + grp) & 7)) * (YK2_RUN * 4) + (slot - grp * (YK2_RUN * 4));
    int L = unit >> 2; const int wave = unit & 3;                            // block (row-major, frames back to back) and strip inside the block
    if (L >= nB) return;
    // batch: per-image pointers of this block's frame (wave-uniform scalars; the kernel argument itself is never copied)
    size_t fr = 0;
    if (P.nFrames > 1) { fr = (size_t)(L / nBf); L -= (int)fr * nBf; }
    const int32_t* const pl0 = P.plane[0] + fr * P.fs.plane; const int32_t* const pl1 = P.plane[1] + fr * P.fs.plane; const int32_t* const pl2 = P.plane[2] + fr * P.fs.plane;
    const uint8_t* const keepP = P.keep ? P.keep + fr * P.fs.keep : nullptr;
    const int32_t* const boundsP = P.bounds ? P.bounds + fr * 16 : nullptr;
    uint8_t* const slotsP = P.slots + fr * P.fs.slots;
    uint32_t* const blockCntP = P.blockCnt + fr * P.fs.blockN;
    const int BY = L / P.xBB64, BX = L - BY * P.xBB64;
    const int w = P.w, h = P.h;

    YK2_PROBE(0);
    if (lane < 24) s_bm[lane] = 0;
    // ---- stage the clamped 65x17 strip (Plane::GetPixelValue clamp, encoder/framework.h:116-121).  All 18 loads of a lane are
    // issued before the first use and none sits in a divergent branch (a branch ends in a wait for the loads it holds: the round
    // trips would run one after the other): addresses are clamped instead, every lane loads, and strips on the image's right edge
    // patch their replicated columns afterwards under a wave-uniform test.  Addresses are a scalar base (first row of the strip)
    // plus a 32-bit byte offset per lane.
    {
        const int g4 = (lane & 15) * 4, r0 = lane >> 4;
        const int gx = BX * 64 + g4;
        const bool inX = gx + 3 < w;                                         // w is a multiple of 8: otherwise gx >= w
        const int gxc = min(gx, w - 4);
        const int gyS = BY * 64 + wave * 16;
        const int gyB = min(gyS, P.hAvail - 1);                              // wave-uniform
        const size_t rowBase = (size_t)gyB * (size_t)P.strideElems;
        const char* const b0 = reinterpret_cast<const char*>(pl0 + rowBase);
        const char* const b1 = reinterpret_cast<const char*>(pl1 + rowBase);
        const char* const b2 = reinterpret_cast<const char*>(pl2 + rowBase);
        int4 R[4], G[4], B[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int rel = min(gyS + r0 + 4 * k, P.hAvail - 1) - gyB;
            const uint32_t off = (uint32_t)(rel * P.strideElems + gxc) * 4u;
            R[k] = *reinterpret_cast<const int4*>(b0 + off);
            G[k] = *reinterpret_cast<const int4*>(b1 + off);
            B[k] = *reinterpret_cast<const int4*>(b2 + off);
        }
        // bottom halo row: the 16 segments are loaded by lanes 0..15 (the other lanes repeat them: same lines, coalesced)
        int4 Rb, Gb, Bb;
        {
            const int rel = min(gyS + 16, P.hAvail - 1) - gyB;
            const uint32_t off = (uint32_t)(rel * P.strideElems + gxc) * 4u;
            Rb = *reinterpret_cast<const int4*>(b0 + off);
            Gb = *reinterpret_cast<const int4*>(b1 + off);
            Bb = *reinterpret_cast<const int4*>(b2 + off);
        }
        // right halo column: 17 samples, lanes 32..48 (the other lanes repeat the first / last one)
        uint32_t hcol;
        const int hr = min(max(lane - 32, 0), 16);
        {
            const int rel = min(gyS + hr, P.hAvail - 1) - gyB;
            const uint32_t off = (uint32_t)(rel * P.strideElems + min(BX * 64 + 64, w - 1)) * 4u;
            hcol = y2_pack3(*reinterpret_cast<const int32_t*>(b0 + off), *reinterpret_cast<const int32_t*>(b1 + off), *reinterpret_cast<const int32_t*>(b2 + off));
        }
        // the reciprocals of the range phase's error terms ride along (every strip pays one load and one LDS store; a coded strip used to
        // compute its 256 quotients itself, four correctly rounded divisions per lane)
        const float4 rc4 = *reinterpret_cast<const float4*>(P.qtab + YK2_QBYTES + (size_t)lane * 16);

        __builtin_amdgcn_s_setprio(0);                                       // all loads are out
        uint4 o[4], ob;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            o[k] = make_uint4(y2_pack3(R[k].x, G[k].x, B[k].x), y2_pack3(R[k].y, G[k].y, B[k].y), y2_pack3(R[k].z, G[k].z, B[k].z), y2_pack3(R[k].w, G[k].w, B[k].w));
            *reinterpret_cast<uint4*>(&s_pix[(r0 + 4 * k) * LS + g4]) = o[k];
        }
        ob = make_uint4(y2_pack3(Rb.x, Gb.x, Bb.x), y2_pack3(Rb.y, Gb.y, Bb.y), y2_pack3(Rb.z, Gb.z, Bb.z), y2_pack3(Rb.w, Gb.w, Bb.w));
        if (lane < 16) *reinterpret_cast<uint4*>(&s_pix[16 * LS + g4]) = ob;
        if (BX * 64 + 64 > w) {                                              // wave-uniform, rare: lanes beyond the right edge replicate column w - 1 (their loads were clamped to columns w-4 .. w-1)
            if (!inX) {
#pragma unroll
                for (int k = 0; k < 4; k++) *reinterpret_cast<uint4*>(&s_pix[(r0 + 4 * k) * LS + g4]) = make_uint4(o[k].w, o[k].w, o[k].w, o[k].w);
                if (lane < 16) *reinterpret_cast<uint4*>(&s_pix[16 * LS + g4]) = make_uint4(ob.w, ob.w, ob.w, ob.w);
            }
        }
        if (lane >= 32 && lane <= 48) s_pix[hr * LS + 64] = hcol;
        *reinterpret_cast<float4*>(&s_rcp[lane * 4]) = rc4;

    }
    __syncthreads();                                                         // single-wave workgroup: an LDS fence
    YK2_PROBE(1);

    // ---- lane geometry: wave = macro-tile row `wave` of the block, lane = macroTile(q)*16 + Morton(cellX, cellY) ------------
    const int q = lane >> 4, cx = y2_cell_x(lane & 15), cy = y2_cell_y(lane & 15);
    const int bxCell = q * 16 + cx * 4, byCell = wave * 16 + cy * 4;          // cell origin inside the block
    const int gxCell = BX * 64 + bxCell, gyCell = BY * 64 + byCell;          // stripe-local pixels
    const int lcell = (cy * 4) * LS + bxCell;                                // strip-local LDS word of the cell origin
    const int lat = cy * 17 + q * 4 + cx;                                    // lattice index of the cell origin
    const bool mtIn = (BX * 64 + q * 16 < w) && (BY * 64 + wave * 16 < h);    // macro-tile origin inside the image

    uint32_t pw[16];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const uint4 t = *reinterpret_cast<const uint4*>(&s_pix[lcell + r * LS]);
        pw[r * 4 + 0] = t.x; pw[r * 4 + 1] = t.y; pw[r * 4 + 2] = t.z; pw[r * 4 + 3] = t.w;
    }

    // ---- a6: seven passes --------------------------------------------------------------------------------------
    unsigned long long cov = 0ULL;                                           // bit = lane = 4x4 cell covered
    {
        // S' is linear in x inside any tile, so |c(x-1)-2c(x)+c(x+1)| <= 4*rejectFactor+1 is necessary for acceptance of every
        // tile containing the three pixels (see tests/csrc/yk_encode_v1.hip).  Row 0 of the cell is tested: six tests already leave a cell of
        // noise alive with probability < 1e-7, a second row only costs the other content instructions.  Packed: channels 0 | 2 of a pixel
        // in the halves of a register, channel 1 of two neighbouring pixels; d + lim as u16 exceeds 2 lim when |d| > lim.
        bool dead = !mtIn;
        {
            const int lim = 4 * P.rejectFactor + 1;
            const y2u2 rb0 = y2_u2(pw[0] & 0x00FF00FFu), rb1 = y2_u2(pw[1] & 0x00FF00FFu), rb2 = y2_u2(pw[2] & 0x00FF00FFu), rb3 = y2_u2(pw[3] & 0x00FF00FFu);
            const y2u2 ga = y2_u2(__builtin_amdgcn_perm(pw[1], pw[0], 0x0C050C01u)), gb = y2_u2(__builtin_amdgcn_perm(pw[2], pw[1], 0x0C050C01u)),
                       gc = y2_u2(__builtin_amdgcn_perm(pw[3], pw[2], 0x0C050C01u));
            const y2u2 l2 = y2_splat(lim);
            const y2u2 u1 = rb0 + rb2 - rb1 - rb1 + l2, u2 = rb1 + rb3 - rb2 - rb2 + l2, ug = ga + gc - gb - gb + l2;
            const y2u2 um = __builtin_elementwise_max(__builtin_elementwise_max(u1, u2), ug);
            dead |= __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(um, y2_splat(2 * lim))) != 0u;
        }
        const unsigned long long deadLanes = __ballot(dead);
        YK2_STAT(75, 1); YK2_STAT(76, ~deadLanes == 0ULL ? 1 : 0); YK2_STAT(77, __popcll(deadLanes));
        const bool stripInside = (BX * 64 + 64 <= w) && (BY * 64 + wave * 16 + 16 <= h);     // no tile of the strip crosses the image's edge
        if (~deadLanes != 0ULL && !YK2_ABLATE(2)) {
            // corner lattice (every 4th pixel, 17 x 5 points incl. the halo): Round6 / Round6P of the three channels at once (SWAR)
            // and the five packed streams of y2_grad_pass
            auto latticePoint = [&](const int idx) {
                const int lr = (idx * 241) >> 12, lc = idx - lr * 17;        // idx / 17 for idx < 85
                const uint32_t raw = s_pix[(lr * 4) * LS + lc * 4];
                const uint32_t r6 = (raw & 0x00FCFCFCu) | ((raw >> 6) & 0x00030303u);                    // EncoderContext.cpp:3183
                const uint32_t z = (raw & 0x007F7F7Fu) + 0x00010101u;
                const uint32_t inc = (z ^ (raw & 0x00808080u)) | (((z & raw & 0x00808080u) >> 7) * 255u);   // min(v + 1, 255) per byte
                const uint32_t p6 = (inc & 0x00FCFCFCu) | ((inc >> 6) & 0x00030303u);                    // EncoderContext.cpp:3202
                s_lat[0 * YK2_LATN + idx] = __builtin_amdgcn_perm(r6, raw, 0x0C040C00u);
                s_lat[1 * YK2_LATN + idx] = __builtin_amdgcn_perm(r6, raw, 0x0C050C01u);
                s_lat[2 * YK2_LATN + idx] = __builtin_amdgcn_perm(r6, raw, 0x0C060C02u);
                s_lat[3 * YK2_LATN + idx] = __builtin_amdgcn_perm(p6, p6, 0x0C010C00u);
                s_lat[4 * YK2_LATN + idx] = __builtin_amdgcn_perm(p6, p6, 0x0C020C02u);
            };
            latticePoint(lane);
            if (lane < YK2_LATN - 64) latticePoint(lane + 64);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 16; k++) pw[k] ^= 0x00808080u;                // bias of the packed passes (bytes - 128)
            if (deadLanes == 0ULL) y2_grad_pass<4, 4, true>(s_lat, lat, cx, cy, pw, cov, deadLanes, stripInside, gxCell, gyCell, w, h, P.rejectFactor, s_bm, bxCell, byCell, s_pix, s_list, s_range, lane);
            else y2_grad_pass<4, 4>(s_lat, lat, cx, cy, pw, cov, deadLanes, stripInside, gxCell, gyCell, w, h, P.rejectFactor, s_bm, bxCell, byCell, s_pix, s_list, s_range, lane);
            if (~(cov | deadLanes) != 0ULL) {
                y2_grad_pass<4, 3>(s_lat, lat, cx, cy, pw, cov, deadLanes, stripInside, gxCell, gyCell, w, h, P.rejectFactor, s_bm, bxCell, byCell, s_pix, s_list, s_range, lane);
                y2_grad_pass<3, 4>(s_lat, lat, cx, cy, pw, cov, deadLanes, stripInside, gxCell, gyCell, w, h, P.rejectFactor, s_bm, bxCell, byCell, s_pix, s_list, s_range, lane);
                y2_grad_pass<3, 3>(s_lat, lat, cx, cy, pw, cov, deadLanes, stripInside, gxCell, gyCell, w, h, P.rejectFactor, s_bm, bxCell, byCell, s_pix, s_list, s_range, lane);
                y2_grad_pass<3, 2>(s_lat, lat, cx, cy, pw, cov, deadLanes, stripInside, gxCell, gyCell, w, h, P.rejectFactor, s_bm, bxCell, byCell, s_pix, s_list, s_range, lane);
                y2_grad_pass<2, 3>(s_lat, lat, cx, cy, pw, cov, deadLanes, stripInside, gxCell, gyCell, w, h, P.rejectFactor, s_bm, bxCell, byCell, s_pix, s_list, s_range, lane);
                y2_grad_pass<2, 2>(s_lat, lat, cx, cy, pw, cov, deadLanes, stripInside, gxCell, gyCell, w, h, P.rejectFactor, s_bm, bxCell, byCell, s_pix, s_list, s_range, lane);
            }
#pragma unroll
            for (int k = 0; k < 16; k++) pw[k] ^= 0x00808080u;
        }
    }
    const int mtIdx = ((BY * 64 + wave * 16) >> 4) * P.mtW + ((BX * 64 + q * 16) >> 4);
    // ---- the strip's small outputs: its share of the seven swizzled bitmaps (word index = swizzle-block index, :3801-3805; every pass packs the
    // strip's tiles into whole bytes of the block's words except 16x16: 4 bits per strip, kept as a byte of their own that yk_scan2_kernel folds
    // into the map), the coverage words of its four macro-tiles and, further down, the sums of its two runs of eight tiles and one 8-byte record
    // per tile.  They all live in ONE allocation (P.small), so that lanes holding different outputs share a store instruction: the scalar base is
    // common, the 32-bit byte offset is the lane's own.  Round 3 issued ~13 single-lane stores and up to five atomics per strip here (7 % of the
    // kernel: profiles/r04/a6_*); now there is one store per access width and no atomic.
    __syncthreads();                                                         // fence: s_bm complete; s_pix dead, its LDS becomes s_lut
    YK2_PROBE(2);
    uint8_t* const small = P.small;
    const uint32_t bmv = s_bm[lane < 24 ? lane : 0];                         // word k of the strip's bitmap image in lane k
    auto bmWord = [&](const int k) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)bmv, k); };
    const uint32_t i64 = (uint32_t)(BY * P.xBB64 + BX);
    const uint32_t frU = (uint32_t)fr;
    uint32_t off32 = ~0u, val32 = 0u;                                        // the 4-byte class is stored further down, with the run sums
    {
        const uint32_t w1 = (uint32_t)wave & 1u, w2 = (uint32_t)wave >> 1;
        // bytes: lane 0 = 16x16 (4 bits), lane 1 = 16x8 (tile rows 2w, 2w+1), lane 2 = 8x16 (tile row w)
        uint32_t off8 = 0u, val8 = 0u;
        off8 = y2_writelane(P.oBm0b + frU * (uint32_t)P.fs.bm0b + i64 * 4u + (uint32_t)wave, 0, off8);
        val8 = y2_writelane((bmWord(0) >> (4 * wave)) & 0xFu, 0, val8);
        off8 = y2_writelane(P.oBm[1] + frU * (uint32_t)P.fs.bitmap[1] + i64 * 4u + (uint32_t)wave, 1, off8);
        val8 = y2_writelane(bmWord(1) >> (8 * wave), 1, val8);
        off8 = y2_writelane(P.oBm[2] + frU * (uint32_t)P.fs.bitmap[2] + i64 * 4u + (uint32_t)wave, 2, off8);
        val8 = y2_writelane(bmWord(2) >> (8 * wave), 2, val8);
        if (lane < 3) Y2_STORE_IF(8, *(small + off8), (uint8_t)val8);
        // 16-bit words: lanes 0, 16, 32, 48 = coverage of the four macro-tiles (bit = cellY*4 + cellX: row-major, the Morton index with its two
        // middle bits exchanged), lane 1 = 8x8 (tile rows 2w, 2w+1), lanes 2, 3 = 4x8 (32x64 swizzle blocks, tile rows 2w, 2w+1)
        const unsigned long long t = ((cov >> 2) ^ cov) & 0x0C0C0C0C0C0C0C0CULL;
        const unsigned long long covRM = cov ^ t ^ (t << 2);
        uint32_t off16 = ((lane & 15) == 0 && mtIn) ? P.oCov + (frU * (uint32_t)P.fs.coverage + (uint32_t)mtIdx) * 2u : ~0u;
        uint32_t val16 = (uint32_t)((covRM >> (q * 16)) & 0xFFFFULL);
        off16 = y2_writelane(P.oBm[3] + frU * (uint32_t)P.fs.bitmap[3] + (i64 * 4u + (uint32_t)wave) * 2u, 1, off16);
        val16 = y2_writelane(bmWord(3 + (int)w2) >> (16 * w1), 1, val16);
#pragma unroll
        for (int sx = 0; sx < 2; sx++) {
            const bool in = BX * 2 + sx < P.xBB32;
            off16 = y2_writelane(in ? P.oBm[5] + frU * (uint32_t)P.fs.bitmap[5] + ((uint32_t)((BY * P.xBB32 + BX * 2 + sx) * 4 + wave)) * 2u : ~0u, 2 + sx, off16);
            val16 = y2_writelane(bmWord(9 + sx * 2 + (int)w2) >> (16 * w1), 2 + sx, val16);
        }
        if (off16 != ~0u) Y2_STORE_IF(8, *reinterpret_cast<uint16_t*>(small + off16), (uint16_t)val16);
        // optional: the packed pixels of the cells nothing covered, for the live 1-D path (yk_set_pixel_cache): [strip][cell row][lane] 16-byte pieces
        if (P.pixCache != nullptr) {
            const bool cellIn = (gxCell < w) && (gyCell < h);
            if (cellIn && !((cov >> lane) & 1ULL)) {
                const uint32_t strip = (uint32_t)((BY * 4 + wave) * P.xBB64 + BX);
                uint4* dstp = P.pixCache + (size_t)strip * 256 + lane;
#pragma unroll
                for (int r = 0; r < 4; r++) dstp[r * 64] = make_uint4(pw[r * 4 + 0], pw[r * 4 + 1], pw[r * 4 + 2], pw[r * 4 + 3]);
            }
        }
        // 32-bit words: lane 1 = 8x4 (64x32 swizzle blocks, tile rows 4w..4w+3), lanes 2, 3 = 4x4 (32x32 swizzle blocks); lanes 4, 5 = the run sums
        {
            const bool in = BY * 2 + (int)w2 < P.yBB32;
            off32 = y2_writelane(in ? P.oBm[4] + frU * (uint32_t)P.fs.bitmap[4] + ((uint32_t)(((BY * 2 + (int)w2) * P.xBB64 + BX) * 2) + w1) * 4u : ~0u, 1, off32);
            val32 = y2_writelane(bmWord(5 + (int)w2 * 2 + (int)w1), 1, val32);
#pragma unroll
            for (int sx = 0; sx < 2; sx++) {
                const bool in6 = in && (BX * 2 + sx < P.xBB32);
                off32 = y2_writelane(in6 ? P.oBm[6] + frU * (uint32_t)P.fs.bitmap[6] + ((uint32_t)(((BY * 2 + (int)w2) * P.xBB32 + BX * 2 + sx) * 2) + w1) * 4u : ~0u, 2 + sx, off32);
                val32 = y2_writelane(bmWord(13 + ((int)w2 * 2 + sx) * 2 + (int)w1), 2 + sx, val32);
            }
        }
    }
    YK2_PROBE(3);
    // ---- a10-a13: range quantiser; an 8x8 tile = the quad of lanes {l & ~3 .. l | 3} -----------------------------------
    int cxB = 0, cyB = 0, cw = w, chh = P.fullH, discard = 1;                 // constraint box of DynamicTileEncode (:4386-4391)
    if (boundsP) {
        const int b0 = boundsP[0], b1 = boundsP[1], b2 = boundsP[2], b3 = boundsP[3];
        discard = (b0 == 0 && b1 == 0 && b2 == w && b3 == P.fullH) ? 1 : 0;    // bbox == whole image: every reject is discarded (EncoderContext.cpp:1294, :1400-1403)
        cxB = (b0 >> 3) << 3; cyB = (b1 >> 3) << 3;
        cw = (((b2 + 7) >> 3) << 3) - cxB; chh = (((b3 + 7) >> 3) << 3) - cyB;
    }
    const int cxl = cx & 1, cyl = cy & 1;
    const int tgx = gxCell - cxl * 4, tgyl = gyCell - cyl * 4, tgy = tgyl + P.y0;   // tile origin (stripe-local / full-image row)
    const bool tileIn = (tgx + 8 <= w) && (tgyl + 8 <= h);
    // LeftRightOrder over the constraint box incl. its zero-size rule (encoder/framework.h:239-255)
    const bool part = tileIn && tgx >= cxB && tgx < cxB + cw && tgy >= cyB && tgy < cyB + chh && (tgx + 8 <= cw) && (tgy + 8 <= chh);
    const bool keepMT = (keepP == nullptr) || discard || (mtIn && keepP[mtIdx] != 0);
    const bool tileLive = part && keepMT;
    const int l00 = lane & ~3;                                               // lane of the tile's top-left cell
    // uncovered quadrants of the lane's 8x8 tile: bits 0..3 = top-left, top-right, bottom-left, bottom-right (the quad's lanes in order)
    const uint32_t quadN = (uint32_t)((~cov) >> l00) & 15u;
    const int v00 = (int)(quadN & 1u), v01 = (int)((quadN >> 2) & 1u);
    const bool valid = tileLive && ((quadN >> (lane & 3)) & 1u);              // valid = mipmapMask && !smoothMap (Plane.cpp:527)
    const int nTop = __popc(quadN & 3u), nBot = __popc(quadN >> 2);
    const int tileIdx = (tgyl >> 3) * P.tilesW + (tgx >> 3);
    const size_t T8 = (size_t)P.tilesW * P.tilesH;
    const int tw = lane >> 2;                                                // tile index inside the wave (0..15)
    const bool writer = ((lane & 3) == 0) && tileIn;                         // one lane per tile writes count / def
    const unsigned long long validMask = __ballot(valid);
    // wave-uniform: every live tile has all four quadrants to code (then every lane of its quad is valid) -- the wide slot stores below
#ifdef YK2_NO_WIDE_STORE
    const bool fullTiles = false;
#else
    const bool fullTiles = !WANT_DST && __ballot(tileLive && quadN != 15u && quadN != 0u) == 0ULL;
#endif

    // ---- first level of the stream compaction's scan, fused: nibbles and coded tiles per run of 8 consecutive tiles (row-major tile order =
    // LeftRightOrder; the counts are the same for the three planes).  A strip holds two runs; when the tile grid is a multiple of 8 wide a run
    // never straddles a scan block of 1024 tiles, and the strip stores each run's sums (nibbles / 16 | coded tiles << 16) in the run's own word:
    // yk_scan2_kernel adds the 128 words of a block.  (Round 3 added them to the block's two counters with atomics: 128 strips of a tile row on
    // one address, 15-23 % of the kernel on noisy frames.)  Other widths keep the atomics.
    {
        const int n16 = (tileLive && !YK2_ABLATE(1)) ? (nTop + nBot) : 0;   // nibbles / 16 of this tile-plane
        if ((P.tilesW & 7) == 0) {
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const bool mine = writer && ((cy >> 1) == r);
                const unsigned long long b0 = __ballot(mine && (n16 & 1)), b1 = __ballot(mine && (n16 & 2)), b2 = __ballot(mine && (n16 & 4));
                const unsigned long long bd = __ballot(mine && n16 > 0);
                const int row = ((BY * 64 + wave * 16) >> 3) + r;
                off32 = y2_writelane(row < P.tilesH ? P.oRun + (frU * (uint32_t)P.fs.runSums + (uint32_t)(row * (P.tilesW >> 3) + BX)) * 4u : ~0u, 4 + r, off32);
                val32 = y2_writelane((uint32_t)(__popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2)) | ((uint32_t)__popcll(bd) << 16), 4 + r, val32);
            }
        } else if (writer && n16 > 0) {
            Y2_ATOMIC(atomicAdd(&blockCntP[((size_t)tileIdx >> 10) * 2], 16u * (uint32_t)n16), n16);
            Y2_ATOMIC(atomicAdd(&blockCntP[((size_t)tileIdx >> 10) * 2 + 1], 1u), 1);
        }
        if (off32 != ~0u) Y2_STORE_IF(8, *reinterpret_cast<uint32_t*>(small + off32), val32);
    }

    YK2_STAT(70, validMask != 0ULL ? 1 : 0); YK2_STAT(74, __popcll(validMask));
    { const int nv = __popcll(validMask); YK2_STAT(120 + (nv == 0 ? 0 : nv <= 4 ? 1 : nv <= 8 ? 2 : nv <= 16 ? 3 : nv <= 32 ? 4 : nv < 64 ? 5 : 6), 1); (void)nv; }
    // one 8-byte record per tile: the three planes' definition words and the tile's nibble count (the same for the three planes), stored once
    // behind the plane loop by the tile's first lane: 16 lanes x 8 bytes = two runs of 64 contiguous bytes per strip (round 3: six stores)
    uint32_t defs01 = 0u, defs2c = 0u;
    if (validMask == 0ULL || YK2_ABLATE(1)) {
    } else {
        uint32_t* lut = &s_lut[WANT_DST ? tw : 0][0];
        // curve constants for buildLut (test-only reconstruction); fetched here, off the path of the strip's pixel loads
        if (WANT_DST) {
            s_curve[lane >> 4][lane & 15] = c_curve2[lane >> 4][lane & 15];
            if (lane < 32) s_curve[4 + (lane >> 4)][lane & 15] = c_curve2[4 + (lane >> 4)][lane & 15];
        }
        const int j4 = lane & 3;                                             // lane index inside its tile (cellY&1)*2 + (cellX&1)
        // ---- at most four cells to code (a smooth strip's last column of cells next to other content, the usual case along contours): 61 idle
        // lanes would watch them walk sixteen pixels each.  Instead the sixteen lanes of a DPP row take one pixel each of one of those cells
        // (`spread`); the cells' own lanes get the six sums back through LDS and carry on as if they had walked the pixels themselves.
        const int nValidLanes = __popcll(validMask);
        const bool spread = nValidLanes <= 4 && !WANT_DST;
        const int vRank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(validMask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)validMask, 0u));
        uint32_t slotOff[4];                                                 // byte offset of the lane's four nibble rows inside the plane's slot array
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int yIn = cyl * 4 + r;
            const int pos = (yIn < 4) ? (yIn * 4 * nTop + (cxl ? 4 * v00 : 0))
                                      : (16 * nTop + (yIn - 4) * 4 * nBot + (cxl ? 4 * v01 : 0));
            slotOff[r] = (uint32_t)tileIdx * (uint32_t)YK_SLOT + (uint32_t)(pos >> 1);
        }
        // ---- Plane::GetMinMax_Y over the tile (Plane.cpp:489-587), the three planes at once: channels 0 | 2 of a pixel in the halves of a
        // register, channel 1 of two pixels; lanes without valid pixels carry the neutral elements; the tile's value over its quad by DPP.
        int tmn[3], tmx[3];
        {
            y2u2 mnRB = y2_splat(0xFFFF), mxRB = y2_splat(0), mnG = y2_splat(0xFFFF), mxG = y2_splat(0);
            if (valid) {
#pragma unroll
                for (int k = 0; k < 16; k += 2) {
                    const y2u2 a = y2_u2(pw[k] & 0x00FF00FFu), b = y2_u2(pw[k + 1] & 0x00FF00FFu), g = y2_u2(__builtin_amdgcn_perm(pw[k + 1], pw[k], 0x0C050C01u));
                    if (k == 0) { mnRB = __builtin_elementwise_min(a, b); mxRB = __builtin_elementwise_max(a, b); mnG = g; mxG = g; }
                    else {
                        mnRB = __builtin_elementwise_min(__builtin_elementwise_min(mnRB, a), b); mxRB = __builtin_elementwise_max(__builtin_elementwise_max(mxRB, a), b);
                        mnG = __builtin_elementwise_min(mnG, g); mxG = __builtin_elementwise_max(mxG, g);
                    }
                }
            }
            tmn[0] = (int)mnRB.x; tmn[2] = (int)mnRB.y; tmx[0] = (int)mxRB.x; tmx[2] = (int)mxRB.y;
            tmn[1] = min((int)mnG.x, (int)mnG.y); tmx[1] = max((int)mxG.x, (int)mxG.y);
#pragma unroll
            for (int p = 0; p < 3; p++) {
                tmn[p] = min(tmn[p], y2_quad<Y2_QUAD_X>(tmn[p])); tmx[p] = max(tmx[p], y2_quad<Y2_QUAD_X>(tmx[p]));
                tmn[p] = min(tmn[p], y2_quad<Y2_QUAD_Y>(tmn[p])); tmx[p] = max(tmx[p], y2_quad<Y2_QUAD_Y>(tmx[p]));
            }
        }
        // ---- DynamicTile::buildTable's integer part (:625-661) from the tile-definition table: the three planes' entries in one round trip
        const uint32_t* const defTab = reinterpret_cast<const uint32_t*>(P.qtab + YK2_QBYTES + YK2_RCPBYTES);
        uint32_t tdef[3];
#pragma unroll
        for (int p = 0; p < 3; p++) {
            const int none = tmn[p] > 255;                                   // no valid pixel in the tile: (0, 0) (Plane.cpp:583-584)
            const int mn = none ? 0 : tmn[p], mx = none ? 0 : tmx[p];
            const int min_ = min(mn, 224);
            const int d8 = max(mx - min_, 32);
            tdef[p] = defTab[(uint32_t)(min_ * YK2_DEFN1 + d8 - 32)];        // unsigned: scalar base + 32-bit lane offset
            s_range[p * 64 + lane] = (uint32_t)mn | ((uint32_t)mx << 8);        // min | max << 8 of the tile's valid pixels, for the tie check (rare): parked in LDS
        }
        // the quantiser table's rows start 256 rows before the table (the row base of the definition table is biased by +256)
        const uint8_t* const qrows = P.qtab - (size_t)256 * 16;
        if (spread && !YK2_ABLATE(4)) {
            // lane (i, k) = pixel k of the i-th cell to code, the three planes one after the other: ONE row and six products per lane and plane, the
            // sums over the sixteen lanes of the row by DPP; sums and index words wait in LDS for the cell's own lane
            if (valid) s_list[vRank] = (uint8_t)lane;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int spI = lane >> 4, spK = lane & 15;                      // pixel of the cell: row spK >> 2, column spK & 3
            const bool spAct = spI < nValidLanes;
            const int spSrc = (int)s_list[spAct ? spI : 0];
            const int sq = spSrc >> 4, scx = y2_cell_x(spSrc & 15), scy = y2_cell_y(spSrc & 15);
            const uint32_t spx = s_pix[(scy * 4 + (spK >> 2)) * LS + sq * 16 + scx * 4 + (spK & 3)];   // the staged strip is intact until the first index word is parked
#pragma unroll
            for (int p = 0; p < 3; p++) {
                const uint32_t q16s = (uint32_t)__builtin_amdgcn_ds_bpermute(spSrc << 2, (int)((tdef[p] & 0xFFFFu) << 4));   // the row base of the source cell's tile
                float t6[6] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
                if (spAct) {
                    const uint32_t v4 = y2_byte_x4(spx, p);
                    const uint4 row = *reinterpret_cast<const uint4*>(qrows + (size_t)((v4 << 2) + q16s));
                    const float rv = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(s_rcp) + v4);
                    if (!MODE3) { y2_fma_lo(t6[0], row.x, rv); y2_fma_hi(t6[1], row.x, rv); y2_fma_lo(t6[2], row.y, rv); }
                    y2_fma_hi(t6[3], row.y, rv); y2_fma_lo(t6[4], row.z, rv); y2_fma_hi(t6[5], row.z, rv);
                    s_sp[96 + (p * 4 + spI) * 16 + spK] = row.w;
                }
#pragma unroll
                for (int m = kStart; m < 6; m++) {
                    const float tot = y2_row16_sum(t6[m]);
                    if (spK == 0) s_sp[(p * 4 + spI) * 8 + m] = __float_as_uint(tot);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // ---- nearest LUT entry per pixel and mode: ONE 16-byte row of the quantiser table per pixel and plane, all sixteen of a plane in flight at once
        uint4 win[16];
        auto issueRows = [&](const int pl, const int k0, const int k1) {
            if (valid && !spread && !YK2_ABLATE(4)) {
                const uint32_t q16 = (tdef[pl] & 0xFFFFu) << 4;
                {
#pragma unroll
                    for (int k = k0; k < k1; k++) {
                        const uint32_t v4 = y2_byte_x4(pw[k], pl);                   // 4 * value: a quarter of its row offset
                        win[k] = *reinterpret_cast<const uint4*>(qrows + (size_t)((v4 << 2) + q16));
                    }
                }
            }
        };
        issueRows(0, 0, YK2_PREF);
#pragma unroll
        for (int p = 0; p < 3; p++) {
            // ---- nearest LUT entry per pixel and mode: ONE 16-byte row of the quantiser table (yk_qtab_kernel).  Every LUT is
            // BN + K[rangeDecode][mode][i] (the float add never carries into the integer part: checked for every (min, max) by
            // yk_selftest 3), so index and minDiff of a pixel depend only on rangeDecode and v - BN.  The table (0.9 MB) lives in
            // L2 and, for the few rangeDecode values a strip meets, in the CU's vector cache.
            // The reference adds the 64 exact terms minDiff/v SEQUENTIALLY in float (:885) and, walking the modes in order, keeps
            // mode m when err_m <= best (:897).  Any summation order of n <= 64 non-negative floats is within gamma_63 = 3.76e-6
            // (relative) of the exact sum and md*rcp(v) is within 2.5e-7 of the correctly rounded quotient, so a screening sum T
            // (tree order, reciprocal) differs from the reference's sum by < 8e-6 relative.  Each of the reference's
            // comparisons is therefore decided with certainty when the two sums are separated by 2e-5, or tie exactly with
            // identical per-pixel minDiffs (then the reference's sums are identical too: the later mode wins), or are both
            // exactly 0 (all terms 0).  Any other case (rare) flags the tile for exact re-summation in the reference's order.
            const uint32_t qrow16 = (tdef[p] & 0xFFFFu) << 4;               // byte offset of the row of v = 0 behind qrows
            float sm[6] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
            issueRows(p, YK2_PREF, 16);                                       // the rows that were not sent ahead
            if (valid && !spread && !YK2_ABLATE(4)) {
                // The plane's sixteen rows are in flight already (issued before the previous plane's mode selection, see below); the index
                // words wait in LDS (word = pixel * 64 + lane: conflict-free) until the mode is chosen.
                float rvw[YK2_RVWIN];
                // the reciprocal of the pixel value (a table: v_rcp_f32 is a quarter-rate op, 16 per plane add up); its LDS round trip is short: a small window of its own
                auto issueRcp = [&](const int k, float& rv) { rv = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(s_rcp) + y2_byte_x4(pw[k], p)); };
#pragma unroll
                for (int k = 0; k < YK2_RVWIN; k++) issueRcp(k, rvw[k]);
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    __builtin_amdgcn_sched_barrier(0);
                    const uint4 row = win[k]; const float rv = rvw[k % YK2_RVWIN];
                    // sum += minDiff * (1 / v): the f16 halves of the row go straight into the multiply-add (one rounding, like __fmaf_rn on the converted value)
                    if (!MODE3) { y2_fma_lo(sm[0], row.x, rv); y2_fma_hi(sm[1], row.x, rv); y2_fma_lo(sm[2], row.y, rv); }
                    y2_fma_hi(sm[3], row.y, rv); y2_fma_lo(sm[4], row.z, rv); y2_fma_hi(sm[5], row.z, rv);
                    s_iw[k * 64 + lane] = row.w;
                    if (k + YK2_RVWIN < 16) issueRcp(k + YK2_RVWIN, rvw[k % YK2_RVWIN]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (spread && valid && !YK2_ABLATE(4)) {                          // the sums and index words the spread lanes left for this cell
#pragma unroll
                for (int m = kStart; m < 6; m++) sm[m] = __uint_as_float(s_sp[(p * 4 + vRank) * 8 + m]);
#pragma unroll
                for (int k4 = 0; k4 < 4; k4++) {
                    const uint4 iw4 = *reinterpret_cast<const uint4*>(&s_sp[96 + (p * 4 + vRank) * 16 + k4 * 4]);
                    s_iw[(k4 * 4 + 0) * 64 + lane] = iw4.x; s_iw[(k4 * 4 + 1) * 64 + lane] = iw4.y; s_iw[(k4 * 4 + 2) * 64 + lane] = iw4.z; s_iw[(k4 * 4 + 3) * 64 + lane] = iw4.w;
                }
            }
            // the first rows of the next plane leave now: their round trip runs under this plane's mode selection, nibble packing and stores
            if (p < 2) issueRows(p + 1, 0, YK2_PREF);
            // the tile's sums over its quad (DPP operands of the add itself); mode selection: the reference keeps mode m when err_m <= best, i.e.
            // the running minimum with the later mode on ties.  The common case only asks whether two sums are surely ordered; everything
            // about sums that are not (`near`: equal, both zero, or closer than the margin) is worked out below, rarely.
            float tm[6] = { sm[0], sm[1], sm[2], sm[3], sm[4], sm[5] };
            y2_quad_sums<kStart>(tm);
            int bestMode = kStart; float bestT = tm[kStart];
            bool near = false;
#pragma unroll
            for (int m = kStart + 1; m < 6; m++) {
                const float t = tm[m];
                const float lo = fminf(t, bestT), hi = fmaxf(t, bestT);
                near |= !(__fmul_rn(lo, 1.00002f) < hi);                      // not surely ordered (this includes the exact ties)
                bestMode = (t <= bestT) ? m : bestMode;
                bestT = lo;
            }
            bool amb = false;
            uint32_t ties = 0;                                               // bit m: mode m tied with the best so far (bits 8+3m..: that mode)
            if (__ballot(near && tileLive) != 0ULL) {                        // wave-uniform, rare: the walk again, with the reasons
                int bm = kStart; float bt = tm[kStart];
#pragma unroll
                for (int m = kStart + 1; m < 6; m++) {
                    const float t = tm[m];
                    const float lo = fminf(t, bt), hi = fmaxf(t, bt);
                    const bool eq = (t == bt);
                    amb |= !eq && !(__fmul_rn(lo, 1.00002f) < hi);            // neither surely ordered nor exactly equal
                    if (eq && t != 0.0f) ties |= (1u << m) | ((uint32_t)bm << (8 + 3 * m));   // both exactly zero needs no check
                    bm = (t <= bt) ? m : bm;
                    bt = lo;
                }
            }
            // exact ties: the later mode wins when the two modes' minDiffs agree on every valid pixel of the tile (identical sums in the
            // reference too); otherwise the tile is ambiguous.  Flat tiles tie all the time (a smooth region next to a contour): instead of
            // walking the pixels again, the rows of ALL values between the tile's minimum and maximum are compared (at most eight: two per lane
            // of the quad, every mode pair at once by xor); a wider range, or a row in which the two modes differ, sends the tile to the exact path.
            if (__ballot(ties != 0u && tileLive) != 0ULL) {
                YK2_STAT(73, 1);
                const uint32_t tr = s_range[p * 64 + lane];
                const uint32_t mn = tr & 255u, mx = tr >> 8;
                const uint32_t va = min(mn + (uint32_t)j4, mx), vb = min(mn + 4u + (uint32_t)j4, mx);
                const uint4 ra = *reinterpret_cast<const uint4*>(qrows + (size_t)((va << 4) + qrow16));
                const uint4 rb = *reinterpret_cast<const uint4*>(qrows + (size_t)((vb << 4) + qrow16));
                // minDiffs as f16 halves: x = modes 0|1, y = 2|3, z = 4|5; a half of an accumulator is non-zero when its pair of modes differs in some row
                const uint32_t rxa = __builtin_amdgcn_alignbit(ra.x, ra.x, 16), rya = __builtin_amdgcn_alignbit(ra.y, ra.y, 16), rza = __builtin_amdgcn_alignbit(ra.z, ra.z, 16);
                const uint32_t rxb = __builtin_amdgcn_alignbit(rb.x, rb.x, 16), ryb = __builtin_amdgcn_alignbit(rb.y, rb.y, 16), rzb = __builtin_amdgcn_alignbit(rb.z, rb.z, 16);
                uint32_t acc[9] = { (ra.x ^ rxa) | (rb.x ^ rxb),                 // (0,1)
                                    (ra.y ^ rya) | (rb.y ^ ryb),                 // (2,3)
                                    (ra.z ^ rza) | (rb.z ^ rzb),                 // (4,5)
                                    (ra.x ^ ra.y) | (rb.x ^ rb.y),               // low (0,2), high (1,3)
                                    (ra.x ^ ra.z) | (rb.x ^ rb.z),               // low (0,4), high (1,5)
                                    (ra.y ^ ra.z) | (rb.y ^ rb.z),               // low (2,4), high (3,5)
                                    (ra.x ^ rya) | (rb.x ^ ryb),                 // low (0,3), high (1,2)
                                    (ra.x ^ rza) | (rb.x ^ rzb),                 // low (0,5), high (1,4)
                                    (ra.y ^ rza) | (rb.y ^ rzb) };               // low (2,5), high (3,4)
#pragma unroll
                for (int i = 0; i < 9; i++) { acc[i] |= (uint32_t)y2_quad<Y2_QUAD_X>((int)acc[i]); acc[i] |= (uint32_t)y2_quad<Y2_QUAD_Y>((int)acc[i]); }
                bool differ = (mx - mn) > 7u;
                auto pairDiffers = [&](const int i, const int j) -> bool {       // i < j
                    const int lo = i < j ? i : j, hi = i < j ? j : i;
                    if (lo == 0 && hi == 1) return (acc[0] & 0xFFFFu) != 0u;
                    if (lo == 2 && hi == 3) return (acc[1] & 0xFFFFu) != 0u;
                    if (lo == 4 && hi == 5) return (acc[2] & 0xFFFFu) != 0u;
                    if (lo == 0 && hi == 2) return (acc[3] & 0xFFFFu) != 0u;
                    if (lo == 1 && hi == 3) return (acc[3] >> 16) != 0u;
                    if (lo == 0 && hi == 4) return (acc[4] & 0xFFFFu) != 0u;
                    if (lo == 1 && hi == 5) return (acc[4] >> 16) != 0u;
                    if (lo == 2 && hi == 4) return (acc[5] & 0xFFFFu) != 0u;
                    if (lo == 3 && hi == 5) return (acc[5] >> 16) != 0u;
                    if (lo == 0 && hi == 3) return (acc[6] & 0xFFFFu) != 0u;
                    if (lo == 1 && hi == 2) return (acc[6] >> 16) != 0u;
                    if (lo == 0 && hi == 5) return (acc[7] & 0xFFFFu) != 0u;
                    if (lo == 1 && hi == 4) return (acc[7] >> 16) != 0u;
                    if (lo == 2 && hi == 5) return (acc[8] & 0xFFFFu) != 0u;
                    return (acc[8] >> 16) != 0u;                                 // (3,4)
                };
#pragma unroll
                for (int m = kStart + 1; m < 6; m++) {
                    const uint32_t fb = ((ties >> m) & 1u) ? ((ties >> (8 + 3 * m)) & 7u) : 7u;   // the mode that mode m tied with, 7 = none
#pragma unroll
                    for (int b2 = kStart; b2 < m; b2++) differ |= (fb == (uint32_t)b2) && pairDiffers(b2, m);
                }
                if (ties != 0u && differ) amb = true;
            }
            uint32_t cLo = 0, cHi = 0;
            if (valid) {                                                     // the lane's 16 index nibbles of the best mode, pixel 0 lowest
                const uint32_t sh = 4u * (uint32_t)bestMode;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    // pixel k's index nibble enters at the top and shifts down: after 8 pixels pixel 0 sits lowest
                    const uint32_t code = s_iw[k * 64 + lane] >> sh;
                    if (k < 8) cLo = __builtin_amdgcn_alignbit(code, cLo, 4); else cHi = __builtin_amdgcn_alignbit(code, cHi, 4);
                }
            }
            if (YK2_ABLATE(16)) amb = true;                                   // test hook: force the exact re-summation everywhere
            unsigned long long ambMask = __ballot(amb && tileLive);
            YK2_STAT(72, __popcll(ambMask) / 4); YK2_STAT(71, ambMask != 0ULL ? 1 : 0);
#ifdef YK2_TIMING
            if (lane == 0 && unit < 65536 && ambMask) { g_y2_times[(size_t)unit * 16 + 10] += (unsigned long long)__popcll(ambMask) / 4; g_y2_times[(size_t)unit * 16 + 11] += 1; }
#endif
            if (WANT_DST) {
                // The tile's six LUTs (4-bit: 16 entries, 3-bit: 8 entries; stored << 8) in LDS, built the reference's way (:662-696); only the
                // test-only reconstruction reads them.  Lane j4 of the tile builds entries 4*j4..4*j4+3 (4-bit) and 2*j4, 2*j4+1 (3-bit).
                const int rowBase = (int)(tdef[p] & 0xFFFFu) - 256 - 2;      // (R - 32) * 258 - BN
                const int base = (int)((tdef[p] >> 16) & 127u);
                const int BN = (base * 224) / 63;
                const int rangeDecode = (rowBase + BN) / YK2_QROWS + YK2_QR0;
                const float Rf = (float)rangeDecode, BNf = (float)BN;
#pragma unroll
                for (int m = 0; m < 3; m++) {
                    uint32_t Lq[4];
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        Lq[k] = (uint32_t)__float2int_rz(__fadd_rn(BNf, __fmul_rn(s_curve[m][j4 * 4 + k], Rf)));
                    *reinterpret_cast<uint4*>(&lut[m * 20 + j4 * 4]) = make_uint4(Lq[0] << 8, Lq[1] << 8, Lq[2] << 8, Lq[3] << 8);
                }
#pragma unroll
                for (int m = 3; m < 6; m++) {
                    uint32_t Lq[2];
#pragma unroll
                    for (int k = 0; k < 2; k++)
                        Lq[k] = (uint32_t)__float2int_rz(__fadd_rn(BNf, __fmul_rn(s_curve[m][j4 * 2 + k], Rf)));
                    *reinterpret_cast<uint2*>(&lut[60 + (m - 3) * 8 + j4 * 2]) = make_uint2(Lq[0] << 8, Lq[1] << 8);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            while (ambMask != 0ULL) {                                        // wave-uniform loop over the ambiguous tiles (rare)
                // The whole wave re-sums ONE tile-plane in the reference's order: the tile's four lanes publish its 64 values, lane i takes
                // pixel i (row-major in the tile): one row gather and six exact quotients per lane, then six lanes add the 64 terms of a mode
                // one after the other.  One memory round trip per tile-plane (a lane walking its own sixteen pixels took sixteen).
                const int a00 = (__ffsll((long long)ambMask) - 1) & ~3;      // top-left lane of that tile
                ambMask &= ~(0xFULL << a00);
                if (l00 == a00) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const uint32_t kSel = 0x0C0C0000u | ((4u + (uint32_t)p) << 8) | (uint32_t)p;         // byte p of two words (p is unrolled: a constant)
                        const uint32_t lo2 = __builtin_amdgcn_perm(pw[r * 4 + 1], pw[r * 4 + 0], kSel), hi2 = __builtin_amdgcn_perm(pw[r * 4 + 3], pw[r * 4 + 2], kSel);
                        s_tile[(cyl * 4 + r) * 2 + cxl] = __builtin_amdgcn_perm(hi2, lo2, 0x05040100u);
                    }
                }
                const uint32_t tq16 = (uint32_t)__builtin_amdgcn_readlane((int)qrow16, a00), tqn = (uint32_t)__builtin_amdgcn_readlane((int)quadN, a00);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                {
                    const uint32_t v = (uint32_t)reinterpret_cast<const uint8_t*>(s_tile)[lane];
                    const bool pv = ((tqn >> (((lane >> 5) << 1) | ((lane >> 2) & 1))) & 1u) != 0u;      // the pixel's quadrant is uncovered (the tile is live)
                    uint4 rw = make_uint4(0u, 0u, 0u, 0u);
                    if (pv) rw = *reinterpret_cast<const uint4*>(qrows + (size_t)((v << 4) + tq16));
                    // minDiff / v as the reference's divss (:885): correctly rounded reciprocal + one correction step (yk_selftest 0);
                    // a skipped pixel (covered or v == 0) contributes +0, which leaves a float sum unchanged
                    const float fv = (float)v, rr = pv ? s_rcp[v] : 0.0f;
                    const y2h2 h01 = __builtin_bit_cast(y2h2, rw.x), h23 = __builtin_bit_cast(y2h2, rw.y), h45 = __builtin_bit_cast(y2h2, rw.z);
                    s_chain[0][lane] = yk_div_exact((float)h01.x, fv, rr);
                    s_chain[1][lane] = yk_div_exact((float)h01.y, fv, rr);
                    s_chain[2][lane] = yk_div_exact((float)h23.x, fv, rr);
                    s_chain[3][lane] = yk_div_exact((float)h23.y, fv, rr);
                    s_chain[4][lane] = yk_div_exact((float)h45.x, fv, rr);
                    s_chain[5][lane] = yk_div_exact((float)h45.y, fv, rr);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (lane < 6 && lane >= kStart) {                            // errorDist += minDiff / v in row-major pixel order (:885)
                    float sacc = 0.0f;
                    const float4* cp = reinterpret_cast<const float4*>(&s_chain[lane][0]);
#pragma unroll
                    for (int k = 0; k < 16; k++) {
                        const float4 a = cp[k];
                        sacc = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(sacc, a.x), a.y), a.z), a.w);
                    }
                    s_err[lane] = sacc;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (l00 == a00) {                                            // last mode whose error is <= the best so far (:897-905)
                    bestMode = -1; float bestErr = 99999999.0f;
                    for (int m = kStart; m < 6; m++) {
                        const float e = s_err[m];
                        if (e <= bestErr) { bestErr = e; bestMode = m; }
                    }
                    if (valid) {
                        const uint32_t sh = 4u * (uint32_t)bestMode;
#pragma unroll
                        for (int k = 0; k < 16; k++) {
                            const uint32_t code = s_iw[k * 64 + lane] >> sh;
                            if (k < 8) cLo = __builtin_amdgcn_alignbit(code, cLo, 4); else cHi = __builtin_amdgcn_alignbit(code, cHi, 4);
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            // codes of the best mode, nibble-packed at the position of the lane's pixels among the tile's valid pixels (:1174-1190)
            if (valid && fullTiles) {
                // every tile of the strip with anything to code is coded whole (no quadrant covered: noise, mild noise, most of a photograph): a tile
                // row is [left cell's 4 nibbles | right cell's 4 nibbles], so the left cell's lane takes its neighbour's two words (DPP) and stores the
                // 16 bytes of its half of the slot at once -- a wave then writes two runs of 256 contiguous bytes per plane instead of 256 scattered
                // 2-byte pieces (the stores were 4.5 % of the frame's kernel time and 28 % of an all-noise frame's: profiles/r04/a2_*)
                uint8_t* const slotPlane = slotsP + (size_t)p * T8 * YK_SLOT;
                const uint32_t nLo = (uint32_t)y2_quad<Y2_QUAD_X>((int)cLo), nHi = (uint32_t)y2_quad<Y2_QUAD_X>((int)cHi);
                if (cxl == 0) {
                    uint32_t so = slotOff[0];
                    asm volatile("" : "+v"(so));
                    const uint4 half = make_uint4(__builtin_amdgcn_perm(nLo, cLo, 0x05040100u), __builtin_amdgcn_perm(nLo, cLo, 0x07060302u),
                                                  __builtin_amdgcn_perm(nHi, cHi, 0x05040100u), __builtin_amdgcn_perm(nHi, cHi, 0x07060302u));
                    if (YK2_NOSTORE & 1) Y2_SINK(half.x ^ half.y ^ half.z ^ half.w); else *reinterpret_cast<uint4*>(slotPlane + so) = half;
                }
            } else if (valid) {
                uint8_t* const slotPlane = slotsP + (size_t)p * T8 * YK_SLOT;      // wave-uniform base + 32-bit lane offsets (3 * T8 * 32 < 2^32)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t code16 = ((r < 2 ? cLo : cHi) >> (16 * (r & 1))) & 0xFFFFu;
                    // the offsets do not depend on the plane; kept as four 32-bit registers and widened here, next to the store, so that
                    // the store takes the scalar base + 32-bit offset form (hoisted 64-bit offsets cost 8 registers and a spill)
                    uint32_t so = slotOff[r];
                    asm volatile("" : "+v"(so));
                    Y2_STORE_IF(1, *reinterpret_cast<uint16_t*>(slotPlane + so), (uint16_t)code16);
                    if (WANT_DST) {
                        const uint32_t* lb = lut + (bestMode < 3 ? bestMode * 20 : 60 + (bestMode - 3) * 8);
                        int32_t* drow = P.dst[p] + (uint32_t)((gyCell + r) * w + gxCell);     // 32-bit element offset from a scalar base (w, h <= 32760)
#pragma unroll
                        for (int i = 0; i < 4; i++) drow[i] = (int32_t)(lb[(code16 >> (4 * i)) & 15u] >> 8);
                    }
                }
            }
            {
                // TileInfo fields are u8 (:506-515); EncodeTileType(type,range,base) (include/YAIK_private.h:358) as u16
                const uint32_t def16 = (((uint32_t)bestMode & 255u) << 13) | (tdef[p] >> 16);
                if (p == 0) defs01 = def16 & 0xFFFFu; else if (p == 1) defs01 |= def16 << 16; else defs2c = def16 & 0xFFFFu;
            }
            __builtin_amdgcn_wave_barrier();
        }
        defs2c |= (uint32_t)(tileLive ? 16 * (nTop + nBot) : 0) << 16;
    }
    if (writer) {
        const uint32_t offI = P.oInfo + (frU * (uint32_t)P.fs.tileInfo + (uint32_t)tileIdx) * 8u;      // scalar base + 32-bit lane offset
        Y2_STORE(*reinterpret_cast<uint2*>(small + offI), make_uint2(defs01, defs2c));
    }
    YK2_PROBE(4);
#if YK2_NOSTORE
    if (y2sink == 0x9E3779B9u) P.coverage[lane] = 1;
#endif
    };   // strip
    strip((int)blockIdx.x >> 3, (int)blockIdx.x & 7, lane, (Y2ParamPtr)__builtin_amdgcn_kernarg_segment_ptr());
}

// one set of tables per device and process, built on first use: quantiser rows | reciprocals | tile definitions
static uint8_t* g_qtab[64] = {};
static std::mutex g_qtabMu;
int yk_qtab_get(yk_ctx* c) {
    std::lock_guard<std::mutex> guard(g_qtabMu);
    if (c->device < 0 || c->device >= 64) return yk_fail(c, YK_ERR_BAD_ARG, "device index");
    if (!g_qtab[c->device]) {
        uint8_t* t = nullptr;
        // 4 KB in front: the kernel addresses the rows from 256 rows before the table (row bases are biased by +256)
        YK_HIP(c, hipMalloc(&t, 4096 + YK2_TABBYTES));
        YK_HIP(c, hipMemsetAsync(t, 0, 4096, c->stream));
        t += 4096;
        hipLaunchKernelGGL(yk_qtab_kernel, dim3(YK2_QNR), dim3(320), 0, c->stream, reinterpret_cast<uint4*>(t));
        {   // behind the rows: RN(1 / v) for v = 1..255, [0] = 0 (a skipped term, :884); IEEE division on the host = __fdiv_rn
            float rcp[256]; rcp[0] = 0.0f;
            for (int v = 1; v < 256; v++) rcp[v] = 1.0f / (float)v;
            YK_HIP(c, hipMemcpyAsync(t + YK2_QBYTES, rcp, sizeof rcp, hipMemcpyHostToDevice, c->stream));
        }
        hipLaunchKernelGGL(yk_deftab_kernel, dim3(225), dim3(YK2_DEFN1), 0, c->stream, reinterpret_cast<uint32_t*>(t + YK2_QBYTES + YK2_RCPBYTES));
        YK_HIP(c, hipGetLastError());
        YK_HIP(c, hipStreamSynchronize(c->stream));
        g_qtab[c->device] = t;
    }
    c->qtab = g_qtab[c->device];
    return YK_OK;
}
void yk_selftest_qtab_launch(yk_ctx* c, int* mismatches) {
    hipLaunchKernelGGL(yk_selftest_qtab_kernel, dim3(256), dim3(256), 0, c->stream, reinterpret_cast<const uint4*>(c->qtab),
                       reinterpret_cast<const uint32_t*>(c->qtab + YK2_QBYTES + YK2_RCPBYTES), mismatches);
}

int yk_launch_encode2(yk_ctx* c, const YkEncodeParams& P) {
    if (!P.qtab) return yk_fail(c, YK_ERR_STATE, "quantiser table missing");
    if (P.startMode != 0 && P.startMode != 3) return yk_fail(c, YK_ERR_BAD_ARG, "startMode must be 0 or 3");
    const int nB = P.xBB64 * P.yBB64 * P.nFrames, group = 8 * YK2_RUN;
    // (the 16x16 map needs no clearing any more: the strips store their 4 bits as bytes of their own, yk_scan2_kernel folds them into the map)
    dim3 grid(((nB + group - 1) / group) * group * 4);
    if (P.wantDst) {
        if (P.startMode) hipLaunchKernelGGL((yk_encode2_kernel<true, true>), grid, dim3(64), 0, c->stream, P);
        else hipLaunchKernelGGL((yk_encode2_kernel<true, false>), grid, dim3(64), 0, c->stream, P);
    } else {
        if (P.startMode) hipLaunchKernelGGL((yk_encode2_kernel<false, true>), grid, dim3(64), 0, c->stream, P);
        else hipLaunchKernelGGL((yk_encode2_kernel<false, false>), grid, dim3(64), 0, c->stream, P);
    }
    YK_HIP(c, hipGetLastError());
    return YK_OK;
}
