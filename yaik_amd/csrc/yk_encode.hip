// yk_encode.hip — gfx950 kernels for the YAIK tile-encode hot path.
//
//   yk_alpha_kernel        a9   EncoderContext::MipPrefilter / quadRecursion   (encoder/EncoderContext.cpp:357-430, 1257-1427)
//   yk_encode_kernel       a6   7x EncoderContext::FittingQuadSmooth           (encoder/EncoderContext.cpp:3710-4363)
//                          a10-a13 DynamicTileEncode / GetMinMax_Y / GetTileDynamic_Y / buildTable
//                                                                              (encoder/EncoderContext.cpp:625-1212, 4365-4602; Plane.cpp:489)
//   yk_scan*/yk_pack       stream compaction of the per-tile results into the reference's global streams
//                          (`streamTileDef` :4419, `streamTileIdx` :4421, nibble packing :1180-1184)
//
// Data layout in HBM: inputs are the reference's planar int32 `Plane`s (4 B/sample, read exactly once by the fused
// kernel).  One workgroup (4 wave64) owns one 64x64-pixel block = one swizzle block of the tile bitmaps
// (include/YAIK_private.h:212-276), stages its 65x65 clamped RGB samples in LDS as packed 0x00BBGGRR words, and each
// wave walks four 16x16 macro-tiles: all seven gradient passes, then the range quantiser of the 2x2 8x8 tiles x 3 planes.
// No MFMA: integer / byte work bounded by HBM streaming of 12-16 B/pixel.
#include "yk_common.h"
#include "yk_curves.h"
#include "yk_device.h"

__constant__ float c_curve[6][16] = YK_CURVE_TABLE;

// ------------------------------------------------------------------------------------------------------------------
// a9: alpha tile-reject, stage 1.  keep[mt] = 1 iff any of the 256 alphas of the aligned 16x16 block is non-zero (closed
// form of quadRecursion with maxMipLevel 3, EncoderContext.cpp:394-423); kept blocks grow the bounding box (:416-422).
//
// The plane is streamed in memory order like a reduction: a work unit is (row, 4096-pixel segment), every lane has four
// 16-byte loads in flight and a wave instruction covers 1 KB of one row.  Four adjacent lanes hold the 16 pixels of one
// tile row; a non-zero group raises the tile's flag with an idempotent byte store into the pre-zeroed map, so the
// streaming path has no atomics.  The bounding box is derived from the flags afterwards (yk_alpha_bbox_kernel).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void yk_alpha_kernel(const int32_t* __restrict__ alpha0, int strideElems, int w, int h,
                                                       uint8_t* __restrict__ keep0, int mtW, int32_t* __restrict__ bounds,
                                                       int nFrames, unsigned long long planeStride, unsigned long long keepStride) {
    // accumulators of the bounding-box kernel that follows on the stream: {x0,y0,x1,y1} = empty, done-counter = 0 (one set per frame)
    for (int f = blockIdx.x; f < nFrames; f += gridDim.x)
        if (threadIdx.x < 5) bounds[(size_t)f * 16 + 8 + threadIdx.x] = threadIdx.x < 2 ? 9999999 : (threadIdx.x < 4 ? -1 : 0);
    const int lane = threadIdx.x & 63;
    const int vecPerRow = w >> 2;                                // int4 per image row (w is a multiple of 8)
    const int nSeg = (vecPerRow + 1023) >> 10;
    const int nUnits = nSeg * h;
    for (long long uu = blockIdx.x; uu < (long long)nUnits * nFrames; uu += gridDim.x) {
        const int f = (int)(uu / nUnits), u = (int)(uu - (long long)f * nUnits);
        const int32_t* alpha = alpha0 + (size_t)f * planeStride;
        uint8_t* keep = keep0 + (size_t)f * keepStride;
        const int y = u / nSeg, seg = u - y * nSeg;
        const int32_t* row = alpha + (size_t)y * strideElems;
        int4 a[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int xv = seg * 1024 + k * 256 + threadIdx.x;
            a[k] = make_int4(0, 0, 0, 0);
            if (xv < vecPerRow) a[k] = *reinterpret_cast<const int4*>(row + xv * 4);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int xv = seg * 1024 + k * 256 + threadIdx.x;
            const unsigned long long b = __ballot((a[k].x | a[k].y | a[k].z | a[k].w) != 0);
            if ((lane & 3) == 0 && ((b >> lane) & 0xFULL) != 0) keep[(size_t)(y >> 4) * mtW + (xv >> 2)] = 1;
        }
    }
}

// bounding box of the kept 16x16 tiles (quadRecursion's boundingL/T/R/B, EncoderContext.cpp:416-422) from the keep flags;
// one atomic per workgroup and bound (a single address only sustains ~88 atomics/us).  The last workgroup to finish also
// publishes the whole-image form: bounds[0..3] = the box, bounds[4] = "bbox == whole image -> every reject discarded"
// (EncoderContext.cpp:1294, :1400-1403); a stripe caller overrides these five ints with the host-combined box.
__global__ __launch_bounds__(256) void yk_alpha_bbox_kernel(const uint32_t* __restrict__ keep4, int mtW, int mtH, int y0, int32_t* __restrict__ bounds,
                                                            int fullW, int fullH, unsigned long long keepStrideWords) {
    __shared__ int s_red[4][4];
    keep4 += (size_t)blockIdx.y * keepStrideWords;               // blockIdx.y = frame of a batch
    bounds += (size_t)blockIdx.y * 16;
    int32_t* bbox = bounds + 8;
    int x0 = 9999999, x1 = -1, gy0 = 9999999, gy1 = -1;
    const int n = mtW * mtH, n4 = (n + 3) >> 2;                  // four 1-byte flags per load (the map is padded to a multiple of 4)
    for (int i4 = blockIdx.x * blockDim.x + threadIdx.x; i4 < n4; i4 += gridDim.x * blockDim.x) {
        uint32_t f = keep4[i4];
        if (!f) continue;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = i4 * 4 + k;
            if (((f >> (8 * k)) & 255u) && i < n) {
                const int my = i / mtW, mx = i - my * mtW;
                x0 = min(x0, mx * 16); x1 = max(x1, mx * 16 + 16);
                gy0 = min(gy0, y0 + my * 16); gy1 = max(gy1, y0 + my * 16 + 16);
            }
        }
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        x0 = min(x0, __shfl_xor(x0, d)); x1 = max(x1, __shfl_xor(x1, d));
        gy0 = min(gy0, __shfl_xor(gy0, d)); gy1 = max(gy1, __shfl_xor(gy1, d));
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_red[wave][0] = x0; s_red[wave][1] = gy0; s_red[wave][2] = x1; s_red[wave][3] = gy1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; k++) {
            x0 = min(x0, s_red[k][0]); gy0 = min(gy0, s_red[k][1]); x1 = max(x1, s_red[k][2]); gy1 = max(gy1, s_red[k][3]);
        }
        if (x1 >= 0) { atomicMin(&bbox[0], x0); atomicMin(&bbox[1], gy0); atomicMax(&bbox[2], x1); atomicMax(&bbox[3], gy1); }
        __threadfence();
        if (atomicAdd(&bbox[4], 1) == (int)gridDim.x - 1) {       // every other workgroup's atomics are visible now
            const int bx0 = atomicAdd(&bbox[0], 0), by0 = atomicAdd(&bbox[1], 0), bx1 = atomicAdd(&bbox[2], 0), by1 = atomicAdd(&bbox[3], 0);
            bounds[0] = bx0; bounds[1] = by0; bounds[2] = bx1; bounds[3] = by1;
            bounds[4] = (bx0 == 0 && by0 == 0 && bx1 == fullW && by1 == fullH) ? 1 : 0;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// fused gradient fit + range quantiser
// ------------------------------------------------------------------------------------------------------------------
struct LaneGeo {
    int lane, cell, cx, cy, row, px0, py;   // cell = 4x4 cell of the macro-tile, lane owns pixels (px0..px0+3, py)
};

__device__ __forceinline__ int yk_byte(uint32_t w, int ch) { return (w >> (8 * ch)) & 255; }
__device__ __forceinline__ int yk_round6(int v) { return (v & ~3) | (v >> 6); }                       // EncoderContext.cpp:3183
__device__ __forceinline__ int yk_round6p(int v) { v = min(v + 1, 255); return (v & ~3) | (v >> 6); } // EncoderContext.cpp:3202

// One gradient pass over one macro-tile (a wave).  Exact integer reformulation of the test at EncoderContext.cpp:3894-3998:
// all weights are multiples of 64 (:3735-3737), so with 1/16-unit weights S' = (TL*lx+TR*rx)*ty + (BL*lx+BR*rx)*by fits 16 bits,
// blendCO = S'>>8 and blendC = (S'+127)>>8, and |cur-blend| <= 3 for both roundings is a range test on D = S' - 256*cur:
//   no-rounding variant passes  <=>  -768 <= D <= 1023 ;  rounding variant passes  <=>  -895 <= D <= 896   (rejectFactor 3)
template <int SX, int SY>
__device__ __forceinline__ void yk_grad_pass(const uint32_t* s_pix, int lbase, const LaneGeo& g, const int (&cc)[12], unsigned long long& cov,
                                             int gx0, int gy0, int w, int h, int rf, uint32_t* s_bm, int bx0, int by0) {
    constexpr int TX = 1 << SX, TY = 1 << SY;
    const int ox = g.px0 & ~(TX - 1), oy = g.py & ~(TY - 1);
    const int ocell = (oy >> 2) * 4 + (ox >> 2);
    const bool inside = (gx0 + ox + TX <= w) && (gy0 + oy + TY <= h);
    const bool allow = inside && (((cov >> (ocell * 4)) & 1ULL) == 0);     // top-left pixel of the tile uncovered (:3871-3875)
    if (__ballot(allow) == 0ULL) return;

    const int wy = 16 - ((g.py - oy) << (4 - SY)), wb = 16 - wy;
    const int dx0 = g.px0 - ox;
    const uint32_t tl = s_pix[lbase + oy * YK_LSTRIDE + ox], tr = s_pix[lbase + oy * YK_LSTRIDE + ox + TX];
    const uint32_t bl = s_pix[lbase + (oy + TY) * YK_LSTRIDE + ox], br = s_pix[lbase + (oy + TY) * YK_LSTRIDE + ox + TX];
    const int loO = -256 * rf, hiO = 256 * rf + 255, loR = loO - 127, hiR = hiO - 127;

    int fail = 0;
#pragma unroll
    for (int set = 0; set < 3; set++) {
        int mn = 0x7fffffff, mx = -0x7fffffff;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            int a = yk_byte(tl, ch), b = yk_byte(tr, ch), c = yk_byte(bl, ch), d = yk_byte(br, ch);
            if (set == 1) { a = yk_round6(a); b = yk_round6(b); c = yk_round6(c); d = yk_round6(d); }
            if (set == 2) { a = yk_round6p(a); b = yk_round6p(b); c = yk_round6p(c); d = yk_round6p(d); }
            const int L = a * wy + c * wb, R = b * wy + d * wb;
            const int dL = L - R, R16 = R << 4;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int lx = 16 - ((dx0 + i) << (4 - SX));
                const int D = R16 + dL * lx - cc[i * 3 + ch];
                mn = min(mn, D); mx = max(mx, D);
            }
        }
        const int fO = (mn < loO) | (mx > hiO), fR = (mn < loR) | (mx > hiR);
        fail |= (fO | (fR << 1)) << (2 * set);
        if (set == 0) {
            // Round6 moves a corner by at most 3, Round6P by -2..+4 (a convex blend keeps those bounds on S'/256), so a pixel
            // whose raw D is outside [loR-1024, hiO+768] fails all six variants.  If every allowed lane holds such a pixel the
            // other two corner sets cannot change any decision of this wave.
            const bool hopeless = (mn < loR - 1024) | (mx > hiO + 768);
            if (__ballot(allow && !hopeless) == 0ULL) return;
        }
    }
    // OR the 6 sticky reject flags over the lanes of each tile (lane bits: r0 r1 | cx0 cx1 | cy0 cy1)
    fail |= __shfl_xor(fail, 1); fail |= __shfl_xor(fail, 2);
    if (SX >= 3) fail |= __shfl_xor(fail, 4);
    if (SX >= 4) fail |= __shfl_xor(fail, 8);
    if (SY >= 3) fail |= __shfl_xor(fail, 16);
    if (SY >= 4) fail |= __shfl_xor(fail, 32);
    const bool accept = allow && (fail != 63);                                         // :3998
    cov |= __ballot(accept && g.row == 0);                                             // paint coverage (:4029-4037)
    if (accept && g.px0 == ox && g.py == oy) {                                         // one lane per tile sets the bitmap bit (:4026)
        const int tbx = (bx0 + ox) >> SX, tby = (by0 + oy) >> SY;
        int bit;
        if (SX == 4 && SY == 4) bit = 0 + tby * 4 + tbx;
        else if (SX == 4 && SY == 3) bit = 32 + tby * 4 + tbx;
        else if (SX == 3 && SY == 4) bit = 64 + tby * 8 + tbx;
        else if (SX == 3 && SY == 3) bit = 96 + tby * 8 + tbx;
        else if (SX == 3 && SY == 2) bit = 160 + (tby >> 3) * 64 + (tby & 7) * 8 + tbx;           // two 64x32 swizzle blocks stacked
        else if (SX == 2 && SY == 3) bit = 288 + (tbx >> 3) * 64 + tby * 8 + (tbx & 7);            // two 32x64 side by side
        else bit = 416 + ((tby >> 3) * 2 + (tbx >> 3)) * 64 + (tby & 7) * 8 + (tbx & 7);           // four 32x32
        atomicOr(&s_bm[bit >> 5], 1u << (bit & 31));
    }
}

__global__ __launch_bounds__(256) void yk_encode_kernel(const YkEncodeParams P) {
    __shared__ __attribute__((aligned(16))) uint32_t s_pix[YK_LROWS * YK_LSTRIDE];
    __shared__ uint32_t s_bm[24];
    __shared__ __attribute__((aligned(16))) uint16_t s_lut[4][4][80];
    __shared__ __attribute__((aligned(16))) float s_chain[4][24][68];
    __shared__ __attribute__((aligned(16))) float s_err[4][24];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int BX = blockIdx.x, BY = blockIdx.y;
    const int w = P.w, h = P.h;

    // ---- stage the clamped 65x65 block (Plane::GetPixelValue clamp, encoder/framework.h:116-121) -----------------
    if (tid < 24) s_bm[tid] = 0;
    {
        // all global loads of the block are issued before the first one is consumed (12 x 16 B + halo per thread in flight)
        const int g4 = (tid & 15) * 4, r0 = tid >> 4;
        const int gx = BX * 64 + g4;
        const bool inX = gx + 3 < w;
        int4 R[4], G[4], B[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int gy = min(BY * 64 + r0 + 16 * k, P.hAvail - 1);
            if (inX) {
                const size_t off = (size_t)gy * P.strideElems + gx;
                R[k] = *reinterpret_cast<const int4*>(P.plane[0] + off);
                G[k] = *reinterpret_cast<const int4*>(P.plane[1] + off);
                B[k] = *reinterpret_cast<const int4*>(P.plane[2] + off);
            } else {
                const size_t off = (size_t)gy * P.strideElems + (w - 1);
                const int r = P.plane[0][off], gg = P.plane[1][off], b = P.plane[2][off];
                R[k] = make_int4(r, r, r, r); G[k] = make_int4(gg, gg, gg, gg); B[k] = make_int4(b, b, b, b);
            }
        }
        // bottom halo row (LDS row 64): threads 0..15; right halo column: threads 64..128
        int4 Rb = make_int4(0, 0, 0, 0), Gb = Rb, Bb = Rb;
        if (tid < 16) {
            const int gy = min(BY * 64 + 64, P.hAvail - 1);
            if (inX) {
                const size_t off = (size_t)gy * P.strideElems + gx;
                Rb = *reinterpret_cast<const int4*>(P.plane[0] + off);
                Gb = *reinterpret_cast<const int4*>(P.plane[1] + off);
                Bb = *reinterpret_cast<const int4*>(P.plane[2] + off);
            } else {
                const size_t off = (size_t)gy * P.strideElems + (w - 1);
                const int r = P.plane[0][off], gg = P.plane[1][off], b = P.plane[2][off];
                Rb = make_int4(r, r, r, r); Gb = make_int4(gg, gg, gg, gg); Bb = make_int4(b, b, b, b);
            }
        }
        uint32_t hcol = 0;
        const int hr = tid - 64;
        if (hr >= 0 && hr < YK_LROWS) {
            const int gy = min(BY * 64 + hr, P.hAvail - 1), gxh = min(BX * 64 + 64, w - 1);
            const size_t off = (size_t)gy * P.strideElems + gxh;
            hcol = (uint32_t)P.plane[0][off] | ((uint32_t)P.plane[1][off] << 8) | ((uint32_t)P.plane[2][off] << 16);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint4 o = make_uint4((uint32_t)R[k].x | ((uint32_t)G[k].x << 8) | ((uint32_t)B[k].x << 16),
                                       (uint32_t)R[k].y | ((uint32_t)G[k].y << 8) | ((uint32_t)B[k].y << 16),
                                       (uint32_t)R[k].z | ((uint32_t)G[k].z << 8) | ((uint32_t)B[k].z << 16),
                                       (uint32_t)R[k].w | ((uint32_t)G[k].w << 8) | ((uint32_t)B[k].w << 16));
            *reinterpret_cast<uint4*>(&s_pix[(r0 + 16 * k) * YK_LSTRIDE + g4]) = o;
        }
        if (tid < 16) {
            const uint4 o = make_uint4((uint32_t)Rb.x | ((uint32_t)Gb.x << 8) | ((uint32_t)Bb.x << 16),
                                       (uint32_t)Rb.y | ((uint32_t)Gb.y << 8) | ((uint32_t)Bb.y << 16),
                                       (uint32_t)Rb.z | ((uint32_t)Gb.z << 8) | ((uint32_t)Bb.z << 16),
                                       (uint32_t)Rb.w | ((uint32_t)Gb.w << 8) | ((uint32_t)Bb.w << 16));
            *reinterpret_cast<uint4*>(&s_pix[64 * YK_LSTRIDE + g4]) = o;
        }
        if (hr >= 0 && hr < YK_LROWS) s_pix[hr * YK_LSTRIDE + 64] = hcol;
    }
    __syncthreads();

    LaneGeo g;
    g.lane = lane; g.cell = lane >> 2; g.cx = g.cell & 3; g.cy = g.cell >> 2; g.row = lane & 3;
    g.px0 = g.cx * 4; g.py = g.cy * 4 + g.row;
    const int jt = ((g.cy & 1) * 2 + (g.cx & 1)) * 4 + g.row;         // lane index inside its 8x8 tile (0..15)
    const int t8 = (g.cy >> 1) * 2 + (g.cx >> 1);                     // which of the 2x2 8x8 tiles
    float crv[6];
#pragma unroll
    for (int m = 0; m < 6; m++) crv[m] = c_curve[m][m < 3 ? jt : (jt & 7)];

    // constraint box of DynamicTileEncode (:4386-4391) in full-image pixels
    int cxB = 0, cyB = 0, cw = w, chh = P.fullH, discard = 1;
    if (P.bounds) {
        const int b0 = P.bounds[0], b1 = P.bounds[1], b2 = P.bounds[2], b3 = P.bounds[3];
        discard = P.bounds[4];
        cxB = (b0 >> 3) << 3; cyB = (b1 >> 3) << 3;
        cw = (((b2 + 7) >> 3) << 3) - cxB; chh = (((b3 + 7) >> 3) << 3) - cyB;
    }

    for (int q = 0; q < 4; q++) {
        const int mtxl = q, mtyl = wave;                  // macro-tile inside the block
        const int bx0 = mtxl * 16, by0 = mtyl * 16;
        const int gx0 = BX * 64 + bx0, gy0 = BY * 64 + by0;      // stripe-local pixel origin of the macro-tile
        if (gx0 >= w || gy0 >= h) continue;                // wave-uniform
        const int lbase = by0 * YK_LSTRIDE + bx0;
        const uint4 pw = *reinterpret_cast<const uint4*>(&s_pix[lbase + g.py * YK_LSTRIDE + g.px0]);
        const uint32_t pix[4] = { pw.x, pw.y, pw.z, pw.w };
        int cc[12];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int ch = 0; ch < 3; ch++) cc[i * 3 + ch] = yk_byte(pix[i], ch) << 8;

        // ---- a6: the seven passes in the shipped order (EncoderContext.cpp:9057-9093) -------------------------
        unsigned long long cov = 0ULL;                    // bit 4*cell = cell covered (mapSmoothTile != 0)
        // Necessary condition shared by all passes and all six variants: inside any tile the blend numerator S' is LINEAR in x,
        // so for three horizontally adjacent pixels of one tile |b(x-1) - 2 b(x) + b(x+1)| <= 1 (floor effects), hence an
        // accepted tile needs |c(x-1) - 2 c(x) + c(x+1)| <= 4*rejectFactor + 1 on every channel.  A lane's four pixels always
        // lie in one tile, so a lane violating it kills every tile it belongs to; if all 64 lanes do, no pass can accept.
        bool dead = false;
        {
            const int lim = 4 * P.rejectFactor + 1;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const int c0 = cc[ch] >> 8, c1 = cc[3 + ch] >> 8, c2 = cc[6 + ch] >> 8, c3 = cc[9 + ch] >> 8;
                dead |= (abs(c0 - 2 * c1 + c2) > lim) | (abs(c1 - 2 * c2 + c3) > lim);
            }
        }
        const bool anyAlive = (__ballot(!dead) != 0ULL) && !(P.ablate & 2);
        if (anyAlive) yk_grad_pass<4, 4>(s_pix, lbase, g, cc, cov, gx0, gy0, w, h, P.rejectFactor, s_bm, bx0, by0);
        if (anyAlive && cov != 0x1111111111111111ULL) {
            yk_grad_pass<4, 3>(s_pix, lbase, g, cc, cov, gx0, gy0, w, h, P.rejectFactor, s_bm, bx0, by0);
            yk_grad_pass<3, 4>(s_pix, lbase, g, cc, cov, gx0, gy0, w, h, P.rejectFactor, s_bm, bx0, by0);
            yk_grad_pass<3, 3>(s_pix, lbase, g, cc, cov, gx0, gy0, w, h, P.rejectFactor, s_bm, bx0, by0);
            yk_grad_pass<3, 2>(s_pix, lbase, g, cc, cov, gx0, gy0, w, h, P.rejectFactor, s_bm, bx0, by0);
            yk_grad_pass<2, 3>(s_pix, lbase, g, cc, cov, gx0, gy0, w, h, P.rejectFactor, s_bm, bx0, by0);
            yk_grad_pass<2, 2>(s_pix, lbase, g, cc, cov, gx0, gy0, w, h, P.rejectFactor, s_bm, bx0, by0);
        }
        const int mtIdx = (gy0 >> 4) * P.mtW + (gx0 >> 4);
        if (lane == 0) {
            unsigned long long c = cov; uint32_t bits = 0;
            for (int k = 0; k < 16; k++) bits |= (uint32_t)((c >> (4 * k)) & 1ULL) << k;
            P.coverage[mtIdx] = (uint16_t)bits;
        }

        // ---- a10-a13: range quantiser of the 2x2 8x8 tiles x 3 planes ------------------------------------------
        const int tgx = gx0 + (g.cx >> 1) * 8, tgyl = gy0 + (g.cy >> 1) * 8;        // tile origin, stripe-local
        const int tgy = tgyl + P.y0;                                                // full-image row
        const bool tileIn = (tgx + 8 <= w) && (tgyl + 8 <= h);
        // LeftRightOrder over the constraint box, including its zero-size rule (encoder/framework.h:239-255):
        // result.w = (x + 8 > constraint.w) ? 0 : 8, likewise h  -> such tiles carry no pixel.
        const bool part = tileIn && tgx >= cxB && tgx < cxB + cw && tgy >= cyB && tgy < cyB + chh && (tgx + 8 <= cw) && (tgy + 8 <= chh);
        const bool keepMT = (P.keep == nullptr) || discard || (P.keep[mtIdx] != 0);
        // validity of the four cells of this lane's tile: valid = mipmapMask && !smoothMap (Plane.cpp:527)
        const int c00 = ((g.cy & ~1) * 4 + (g.cx & ~1));
        const bool v00 = !((cov >> (4 * c00)) & 1ULL), v10 = !((cov >> (4 * (c00 + 1))) & 1ULL);
        const bool v01 = !((cov >> (4 * (c00 + 4))) & 1ULL), v11 = !((cov >> (4 * (c00 + 5))) & 1ULL);
        const bool tileLive = part && keepMT;
        const bool valid = tileLive && !((cov >> (4 * g.cell)) & 1ULL);
        const int nTop = (int)v00 + (int)v10, nBot = (int)v01 + (int)v11;
        const int valueCount = tileLive ? 16 * (nTop + nBot) : 0;
        const int tileIdx = (tgyl >> 3) * P.tilesW + (tgx >> 3);
        const size_t T8 = (size_t)P.tilesW * P.tilesH;
        const unsigned long long anyValid = __ballot(valid);

        if (anyValid == 0ULL || (P.ablate & 1)) {
            if (jt == 0 && tileIn) {
#pragma unroll
                for (int p = 0; p < 3; p++) P.tileCount[p * T8 + tileIdx] = 0;
            }
            continue;
        }
        // nibble position of this lane's 4 pixels among the valid pixels of the tile, row-major (:1174-1190)
        const int yIn = (g.cy & 1) * 4 + g.row, xc = g.cx & 1;
        const int pos = (yIn < 4) ? (yIn * 4 * nTop + (xc ? 4 * (int)v00 : 0))
                                  : (16 * nTop + (yIn - 4) * 4 * nBot + (xc ? 4 * (int)v01 : 0));
        const int p0 = yIn * 8 + xc * 4;                 // row-major pixel index inside the tile

        for (int p = 0; p < 3; p++) {
            int v[4];
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = yk_byte(pix[i], p);
            // Plane::GetMinMax_Y over the tile (Plane.cpp:489-587)
            int mn = valid ? min(min(v[0], v[1]), min(v[2], v[3])) : 99999999;
            int mx = valid ? max(max(v[0], v[1]), max(v[2], v[3])) : -99999999;
            mn = min(mn, __shfl_xor(mn, 1)); mx = max(mx, __shfl_xor(mx, 1));
            mn = min(mn, __shfl_xor(mn, 2)); mx = max(mx, __shfl_xor(mx, 2));
            mn = min(mn, __shfl_xor(mn, 4)); mx = max(mx, __shfl_xor(mx, 4));
            mn = min(mn, __shfl_xor(mn, 16)); mx = max(mx, __shfl_xor(mx, 16));
            if (mn == 99999999) { mn = 0; mx = 0; }
            // DynamicTile::buildTable (:625-699)
            const int min_ = min(mn, 224);
            int diff = mx - min_; if (diff < 16) diff = 16;
            const int base = (min_ * 63 + 112) / 224;
            const int BN = (base * 224) / 63;
            const int d8 = max(diff, 32);
            const int scale = 223 - BN;
            // C division by `scale` (truncating).  scale is -1 (base 63) or 3..223 and the numerator is < 2^15, so the quotient
            // is floor((n + 0.5) * rcp(scale)) with a 1-ulp reciprocal: the nearest integer boundary is >= 0.5/223 away
            // while the error is < 0.002 (exhaustively checked by yk_selftest 1).
            const int dnum = (d8 - 32) * 127 + (scale - 1);
            const int dist = (scale < 0) ? -dnum : __float2int_rz(((float)dnum + 0.5f) * __builtin_amdgcn_rcpf((float)scale));
            const int rangeDecode = (dist * scale) / 127 + 32;
            const float Rf = (float)rangeDecode, BNf = (float)BN;
            // lane jt of the tile builds entry jt of every curve; stored pre-shifted (<<4) so that one v_sad_u32 yields
            // (|LUT-v| << 4) + n  and a plain minimum returns the FIRST nearest entry (:873-881)
            uint16_t* lut = &s_lut[wave][t8][0];
#pragma unroll
            for (int m = 0; m < 3; m++) lut[m * 16 + jt] = (uint16_t)(__float2int_rz(__fadd_rn(BNf, __fmul_rn(crv[m], Rf))) << 4);
            if (jt < 8) {
#pragma unroll
                for (int m = 3; m < 6; m++) lut[48 + (m - 3) * 8 + jt] = (uint16_t)(__float2int_rz(__fadd_rn(BNf, __fmul_rn(crv[m], Rf))) << 4);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            uint32_t key[6][4];
            uint32_t vs[4];
#pragma unroll
            for (int i = 0; i < 4; i++) vs[i] = (uint32_t)v[i] << 4;
#pragma unroll
            for (int m = 0; m < 6; m++) {
                const int cnt = m < 3 ? 16 : 8;
                const uint32_t* lw = reinterpret_cast<const uint32_t*>(lut + (m < 3 ? m * 16 : 48 + (m - 3) * 8));
#pragma unroll
                for (int i = 0; i < 4; i++) key[m][i] = 0xFFFFFFFFu;
                if (m >= P.startMode && !(P.ablate & 4)) {
#pragma unroll
                    for (int k = 0; k < cnt / 2; k++) {
                        const uint32_t wv = lw[k];
                        const uint32_t e0 = wv & 0xFFFFu, e1 = wv >> 16;
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            key[m][i] = min(key[m][i], __usad(e0, vs[i], 2 * k));
                            key[m][i] = min(key[m][i], __usad(e1, vs[i], 2 * k + 1));
                        }
                    }
                }
            }
            // per-pixel relative error terms (:884-886), correctly rounded like the reference's divss, handed to the chain lanes
            // through LDS.  One IEEE reciprocal per pixel, then a Markstein step per mode (yk_div_exact; all 256x256 operand
            // pairs are checked against __fdiv_rn on the GPU by yk_selftest / tests/test_gpu_selftest.py).
            float fv[4], rv[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const bool on = valid && v[i] != 0;
                fv[i] = (float)v[i];
                rv[i] = on ? __fdiv_rn(1.0f, fv[i]) : 0.0f;       // r = 0 makes every term exactly +0 (skipped pixel)
            }
#pragma unroll
            for (int m = 0; m < 6; m++) {
                float qv[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float md = (float)(int)(key[m][i] >> 4);
                    qv[i] = (m >= P.startMode) ? yk_div_exact(md, fv[i], rv[i]) : 0.0f;
                }
                *reinterpret_cast<float4*>(&s_chain[wave][t8 * 6 + m][p0]) = make_float4(qv[0], qv[1], qv[2], qv[3]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // sequential float sum in row-major pixel order, one lane per (tile, mode): errorDist += minDiff / v (:885)
            if (lane < 24) {
                float s = 0.0f;
                const float4* cp = reinterpret_cast<const float4*>(&s_chain[wave][lane][0]);
#pragma unroll 4
                for (int k = 0; k < 16; k++) {
                    const float4 a = cp[k];
                    s = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(s, a.x), a.y), a.z), a.w);
                }
                s_err[wave][lane] = s;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // best mode: last mode whose error is <= the best so far (:897-905)
            int bestMode = -1; float bestErr = 99999999.0f;
#pragma unroll
            for (int m = 0; m < 6; m++) {
                const float e = s_err[wave][t8 * 6 + m];
                if (m >= P.startMode && e <= bestErr) { bestErr = e; bestMode = m; }
            }
            uint32_t code4 = 0;
            int code[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                uint32_t k = key[0][i];
#pragma unroll
                for (int m = 1; m < 6; m++) k = (bestMode == m) ? key[m][i] : k;
                code[i] = (int)(k & 15u);
                code4 |= (uint32_t)code[i] << (4 * i);
            }
            if (valid) {
                *reinterpret_cast<uint16_t*>(P.slots + ((size_t)p * T8 + tileIdx) * YK_SLOT + (pos >> 1)) = (uint16_t)code4;
                if (P.wantDst) {
                    const uint16_t* lb = lut + (bestMode < 3 ? bestMode * 16 : 48 + (bestMode - 3) * 8);
                    int32_t* drow = P.dst[p] + (size_t)(gy0 + g.py) * w + gx0 + g.px0;
#pragma unroll
                    for (int i = 0; i < 4; i++) drow[i] = (int32_t)(lb[code[i]] >> 4);
                }
            }
            if (jt == 0 && tileIn) {
                P.tileCount[p * T8 + tileIdx] = (uint8_t)valueCount;
                // TileInfo fields are u8 (:506-515); EncodeTileType(type,range,base) (include/YAIK_private.h:358) stored as u16
                P.tileDef[p * T8 + tileIdx] = (uint16_t)((((uint32_t)bestMode & 255u) << 13) | (((uint32_t)dist & 255u) << 7) | ((uint32_t)base & 255u));
            }
            __builtin_amdgcn_wave_barrier();
        }
    }

    // ---- block bitmaps -> the seven swizzled bitmaps (word index = swizzle-block index, :3801-3805) -----------
    __syncthreads();
    if (tid == 0) {
        const int i64 = BY * P.xBB64 + BX;
        reinterpret_cast<uint16_t*>(P.bitmap[0])[i64] = (uint16_t)s_bm[0];
        reinterpret_cast<uint32_t*>(P.bitmap[1])[i64] = s_bm[1];
        reinterpret_cast<uint32_t*>(P.bitmap[2])[i64] = s_bm[2];
        reinterpret_cast<uint32_t*>(P.bitmap[3])[i64 * 2] = s_bm[3];
        reinterpret_cast<uint32_t*>(P.bitmap[3])[i64 * 2 + 1] = s_bm[4];
        for (int s = 0; s < 2; s++) {
            if (BY * 2 + s < P.yBB32) {                                   // 8x4: 64x32 swizzle blocks
                const int i = (BY * 2 + s) * P.xBB64 + BX;
                reinterpret_cast<uint32_t*>(P.bitmap[4])[i * 2] = s_bm[5 + s * 2];
                reinterpret_cast<uint32_t*>(P.bitmap[4])[i * 2 + 1] = s_bm[6 + s * 2];
            }
            if (BX * 2 + s < P.xBB32) {                                   // 4x8: 32x64 swizzle blocks
                const int i = BY * P.xBB32 + BX * 2 + s;
                reinterpret_cast<uint32_t*>(P.bitmap[5])[i * 2] = s_bm[9 + s * 2];
                reinterpret_cast<uint32_t*>(P.bitmap[5])[i * 2 + 1] = s_bm[10 + s * 2];
            }
        }
        for (int sy = 0; sy < 2; sy++) for (int sx = 0; sx < 2; sx++) {   // 4x4: 32x32 swizzle blocks
            if (BY * 2 + sy < P.yBB32 && BX * 2 + sx < P.xBB32) {
                const int i = (BY * 2 + sy) * P.xBB32 + BX * 2 + sx;
                reinterpret_cast<uint32_t*>(P.bitmap[6])[i * 2] = s_bm[13 + (sy * 2 + sx) * 2];
                reinterpret_cast<uint32_t*>(P.bitmap[6])[i * 2 + 1] = s_bm[14 + (sy * 2 + sx) * 2];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// stream compaction: per-tile (count, def, 32-byte nibble slot) -> the reference's global streams, LeftRightOrder =
// row-major over the tile grid (tiles outside the constraint box carry count 0).
// ------------------------------------------------------------------------------------------------------------------
#define YK_SCAN_TILE 1024

// first scan level for the first-generation kernel (the second-generation kernel accumulates these sums itself)
__global__ __launch_bounds__(1024) void yk_scan1_kernel(const uint8_t* __restrict__ tileCount, size_t T8, uint32_t* __restrict__ blockCnt) {
    __shared__ uint32_t s_tmp[32];
    const size_t i = (size_t)blockIdx.x * YK_SCAN_TILE + threadIdx.x;
    const uint32_t c = (i < T8) ? tileCount[i] : 0;                          // plane 0: the counts do not depend on the plane
    uint32_t tot2, totN;
    yk_block_exscan(c ? 1u : 0u, s_tmp, &tot2);
    yk_block_exscan(c, s_tmp, &totN);
    if (threadIdx.x == 0) { blockCnt[(size_t)blockIdx.x * 2] = totN; blockCnt[(size_t)blockIdx.x * 2 + 1] = tot2; }
}

// second level: exclusive prefix over the per-block sums (consumed and cleared for the next frame) and the totals of the
// three planes (identical: the counts do not depend on the plane).
__global__ __launch_bounds__(1024) void yk_scan2_kernel(uint32_t* __restrict__ blockCnt, uint32_t* __restrict__ blockSums, int nBlocks,
                                                        uint32_t* __restrict__ totals, unsigned long long blockNStride) {
    __shared__ uint32_t s_tmp[32];
    blockCnt += (size_t)blockIdx.x * blockNStride; blockSums += (size_t)blockIdx.x * blockNStride;     // blockIdx.x = frame of a batch
    totals += (size_t)blockIdx.x * 8;
    uint32_t baseN = 0, baseD = 0;
    for (int start = 0; start < nBlocks; start += 1024) {
        const int i = start + threadIdx.x;
        const uint32_t n = i < nBlocks ? blockCnt[i * 2] : 0, d = i < nBlocks ? blockCnt[i * 2 + 1] : 0;
        uint32_t totN, totD;
        const uint32_t en = yk_block_exscan(n, s_tmp, &totN);
        const uint32_t ed = yk_block_exscan(d, s_tmp, &totD);
        if (i < nBlocks) {
            blockSums[i * 2] = baseN + en; blockSums[i * 2 + 1] = baseD + ed;
            blockCnt[i * 2] = 0; blockCnt[i * 2 + 1] = 0;
        }
        baseN += totN; baseD += totD;
    }
    if (threadIdx.x < 3) { totals[threadIdx.x * 2] = baseD; totals[threadIdx.x * 2 + 1] = baseN; }
}

// One workgroup packs the nibbles of 1024 consecutive tiles.  Every tile holds a multiple of 16 nibbles (16 per uncovered 4x4
// quadrant), so every stream offset is a multiple of 8 bytes: after the scan, four lanes per tile copy 8-byte pieces of the
// tile's slot straight to its place in the stream, reading only the bytes that exist.
__global__ __launch_bounds__(1024) void yk_pack_kernel(const uint8_t* __restrict__ tileCount, const uint16_t* __restrict__ tileDef,
                                                       const uint8_t* __restrict__ slots, size_t T8, const uint32_t* __restrict__ blockSums, int nBlocks,
                                                       uint16_t* __restrict__ defsOut, uint32_t* __restrict__ nibOut, size_t nibStrideWords, YkFrameStrides fs) {
    __shared__ uint32_t s_tmp[32];
    __shared__ uint32_t s_off[YK_SCAN_TILE];
    __shared__ uint8_t s_cnt[YK_SCAN_TILE];
    {   // blockIdx.z = frame of a batch
        const size_t f = blockIdx.z;
        tileCount += f * fs.tileCount; tileDef += f * fs.tileDef; slots += f * fs.slots; blockSums += f * fs.blockN;
        defsOut += f * fs.defsOut; nibOut += f * (fs.nibOut / 4);
    }
    const int p = blockIdx.y;
    const size_t i0 = (size_t)blockIdx.x * YK_SCAN_TILE;
    {
        const size_t i = i0 + threadIdx.x;
        const uint32_t c = (i < T8) ? tileCount[p * T8 + i] : 0;
        uint32_t totN, totD;
        const uint32_t en = yk_block_exscan(c, s_tmp, &totN);
        const uint32_t ed = yk_block_exscan(c ? 1u : 0u, s_tmp, &totD);
        const uint32_t baseN = blockSums[(size_t)blockIdx.x * 2], baseD = blockSums[(size_t)blockIdx.x * 2 + 1];     // same for the three planes
        s_off[threadIdx.x] = (baseN + en) >> 1;                                   // byte offset inside the plane's stream
        s_cnt[threadIdx.x] = (uint8_t)c;
        if (c) defsOut[p * T8 + baseD + ed] = tileDef[p * T8 + i];
    }
    __syncthreads();
    uint8_t* out = reinterpret_cast<uint8_t*>(nibOut + (size_t)p * nibStrideWords);
    const int piece = threadIdx.x & 3;
    for (int it = 0; it < 4; it++) {
        const int t = it * 256 + (threadIdx.x >> 2);
        const size_t i = i0 + t;
        if (i >= T8) break;
        if (piece * 8 < (s_cnt[t] >> 1))
            *reinterpret_cast<uint2*>(out + s_off[t] + piece * 8) = *reinterpret_cast<const uint2*>(slots + ((size_t)p * T8 + i) * YK_SLOT + piece * 8);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------------------
int yk_launch_alpha(yk_ctx* c, bool batch) {
    const int F = batch ? c->nFrames : 1;
    uint8_t* keep = batch ? c->B.keep : c->keep;
    int32_t* bounds = batch ? c->B.bounds : c->bounds;
    const int32_t* alpha = batch ? c->B.plane[3] : c->plane[3];
    const size_t nFlags = ((size_t)c->mtW * c->mtH + 3) & ~(size_t)3;
    YK_HIP(c, hipMemsetAsync(keep, 0, F > 1 ? (size_t)c->fs.keep * F : nFlags, c->stream));
    const long long nUnits = (long long)((c->fullW / 4 + 1023) / 1024) * c->h * F;
    hipLaunchKernelGGL(yk_alpha_kernel, dim3((unsigned)(nUnits < 4096 ? nUnits : 4096)), dim3(256), 0, c->stream, alpha, c->strideElems, c->fullW, c->h,
                       keep, c->mtW, bounds, F, (unsigned long long)c->fs.plane, (unsigned long long)c->fs.keep);
    YK_HIP(c, hipGetLastError());
    const int nb = (int)((nFlags / 4 + 1023) / 1024);
    hipLaunchKernelGGL(yk_alpha_bbox_kernel, dim3(nb < 128 ? nb : 128, F), dim3(256), 0, c->stream, reinterpret_cast<const uint32_t*>(keep), c->mtW, c->mtH,
                       c->y0, bounds, c->fullW, c->fullH, (unsigned long long)(c->fs.keep / 4));
    YK_HIP(c, hipGetLastError());
    return YK_OK;
}

int yk_launch_alpha_finish(yk_ctx* c, const int32_t* globalBBox) {
    // whole image: yk_alpha_bbox_kernel already published bounds[0..4]; stripes: the host-combined box replaces them
    if (globalBBox) {
        int32_t b[5] = { globalBBox[0], globalBBox[1], globalBBox[2], globalBBox[3], 0 };
        b[4] = (b[0] == 0 && b[1] == 0 && b[2] == c->fullW && b[3] == c->fullH) ? 1 : 0;
        YK_HIP(c, hipMemcpyAsync(c->bounds, b, sizeof b, hipMemcpyHostToDevice, c->stream));
    }
    return YK_OK;
}

int yk_launch_encode(yk_ctx* c, int rejectFactor, int mode3BitOnly, int wantDst, bool batch) {
    YkEncodeParams P;
    for (int i = 0; i < 4; i++) P.plane[i] = batch ? c->B.plane[i] : c->plane[i];
    P.strideElems = c->strideElems; P.w = c->fullW; P.h = c->h; P.hAvail = c->h + c->halo; P.y0 = c->y0; P.fullH = c->fullH;
    P.rejectFactor = rejectFactor; P.startMode = mode3BitOnly ? 3 : 0; P.wantDst = wantDst; P.ablate = c->ablate;
    P.keep = (c->nPlanes == 4) ? (batch ? c->B.keep : c->keep) : nullptr;
    P.bounds = (c->nPlanes == 4) ? (batch ? c->B.bounds : c->bounds) : nullptr;
    for (int i = 0; i < 7; i++) P.bitmap[i] = batch ? c->B.bitmap[i] : c->bitmap[i];
    P.coverage = batch ? c->B.coverage : c->coverage; P.tileDef = batch ? c->B.tileDef : c->tileDef;
    P.tileCount = batch ? c->B.tileCount : c->tileCount; P.slots = batch ? c->B.slots : c->slots;
    P.blockCnt = c->kernelVersion == 2 ? (batch ? c->B.blockCnt : c->blockCnt) : nullptr;
    for (int i = 0; i < 3; i++) P.dst[i] = c->dst[i];
    P.tilesW = c->tilesW; P.tilesH = c->tilesH; P.mtW = c->mtW; P.mtH = c->mtH;
    P.xBB64 = (c->fullW + 63) / 64; P.yBB64 = (c->h + 63) / 64; P.xBB32 = (c->fullW + 31) / 32; P.yBB32 = (c->h + 31) / 32;
    P.nFrames = batch ? c->nFrames : 1; P.fs = c->fs;
    P.qtab = c->qtab;
    if (c->kernelVersion == 2) return yk_launch_encode2(c, P);
    if (batch) return yk_fail(c, YK_ERR_STATE, "batches need kernel version 2");
    dim3 grid(P.xBB64, P.yBB64);
    hipLaunchKernelGGL(yk_encode_kernel, grid, dim3(256), 0, c->stream, P);
    YK_HIP(c, hipGetLastError());
    return YK_OK;
}

int yk_launch_pack(yk_ctx* c, bool batch) {
    const size_t T8 = (size_t)c->tilesW * c->tilesH;
    const int nb = c->nScanBlocks, F = batch ? c->nFrames : 1;
    uint32_t* blockCnt = batch ? c->B.blockCnt : c->blockCnt; uint32_t* blockSums = batch ? c->B.blockSums : c->blockSums;
    uint32_t* totals = batch ? c->B.totals : c->totals; uint8_t* nibOut = batch ? c->B.nibOut : c->nibOut;
    if (c->kernelVersion != 2) hipLaunchKernelGGL(yk_scan1_kernel, dim3(nb), dim3(1024), 0, c->stream, c->tileCount, T8, c->blockCnt);
    hipLaunchKernelGGL(yk_scan2_kernel, dim3(F), dim3(1024), 0, c->stream, blockCnt, blockSums, nb, totals, (unsigned long long)c->fs.blockN);
    hipLaunchKernelGGL(yk_pack_kernel, dim3(nb, 3, F), dim3(1024), 0, c->stream, batch ? c->B.tileCount : c->tileCount, batch ? c->B.tileDef : c->tileDef,
                       batch ? c->B.slots : c->slots, T8, blockSums, nb, batch ? c->B.defsOut : c->defsOut, reinterpret_cast<uint32_t*>(nibOut), c->nibStride / 4, c->fs);
    YK_HIP(c, hipGetLastError());
    return YK_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// self-test hooks (run by tests/test_gpu_selftest.py): exhaustive checks of the arithmetic shortcuts used above
// ------------------------------------------------------------------------------------------------------------------
__global__ void yk_selftest_div_kernel(int* mismatches) {
    const int n = blockIdx.x, d = threadIdx.x + 1;           // n in 0..255, d in 1..256
    const float fn = (float)n, fd = (float)d;
    const float ref = __fdiv_rn(fn, fd);
    const float got = yk_div_exact(fn, fd, __fdiv_rn(1.0f, fd));
    if (__float_as_uint(ref) != __float_as_uint(got)) atomicAdd(mismatches, 1);
}

__global__ void yk_selftest_scale_kernel(int* mismatches) {
    const int scale = blockIdx.x + 1, d8 = threadIdx.x;        // scale 1..256 (superset of 3..223), d8 32..255
    if (d8 < 32) return;
    const int dnum = (d8 - 32) * 127 + (scale - 1);
    const int ref = dnum / scale;
    const int got = __float2int_rz(((float)dnum + 0.5f) * __builtin_amdgcn_rcpf((float)scale));
    if (ref != got) atomicAdd(mismatches, 1);
}

__global__ void yk_selftest_r1div_kernel(int* mismatches) {
    const int n = blockIdx.x * 16 + (threadIdx.x >> 4), d0 = (threadIdx.x & 15) * 16;     // n 0..4095, d 1..255
    for (int k = 0; k < 16; k++) {
        const int d = d0 + k;
        if (d < 1 || d > 255) continue;
        const int got = __float2int_rz(((float)n + 0.5f) * __builtin_amdgcn_rcpf((float)d));
        if (got != n / d) atomicAdd(mismatches, 1);
    }
}

extern "C" int yk_selftest(yk_ctx* c, int which, int* result) {
    if (!c || !result) return YK_ERR_BAD_ARG;
    YK_HIP(c, hipSetDevice(c->device));
    int* d = nullptr;
    YK_HIP(c, hipMalloc(&d, sizeof(int)));
    YK_HIP(c, hipMemsetAsync(d, 0, sizeof(int), c->stream));
    if (which == 0) hipLaunchKernelGGL(yk_selftest_div_kernel, dim3(256), dim3(256), 0, c->stream, d);
    else if (which == 1) hipLaunchKernelGGL(yk_selftest_scale_kernel, dim3(256), dim3(256), 0, c->stream, d);
    else if (which == 2) hipLaunchKernelGGL(yk_selftest_r1div_kernel, dim3(256), dim3(256), 0, c->stream, d);
    else if (which == 3) yk_selftest_qtab_launch(c, d);
    else { (void)hipFree(d); return yk_fail(c, YK_ERR_BAD_ARG, "unknown selftest"); }
    YK_HIP(c, hipMemcpyAsync(result, d, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    YK_HIP(c, hipStreamSynchronize(c->stream));
    (void)hipFree(d);
    return YK_OK;
}
