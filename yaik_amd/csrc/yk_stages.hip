// yk_stages.hip — the kernels around the fused encode kernel (yk_encode2.hip) and the stage launchers.
//
//   yk_alpha_kernel        a9   EncoderContext::MipPrefilter / quadRecursion   (encoder/EncoderContext.cpp:357-430, 1257-1427)
//   yk_scan*/yk_pack       stream compaction of the per-tile results into the reference's global streams
//                          (`streamTileDef` :4419, `streamTileIdx` :4421, nibble packing :1180-1184)
// The first-generation fused kernel (yk_encode_kernel: lane = one row of four pixels) is no longer part of this library: it lives in
// tests/csrc/yk_encode_v1.hip as an independent second implementation the parity tests cross-check against (yk_set_cross_check_launcher).
//
// No MFMA: integer / byte work bounded by HBM streaming.
#include "yk_common.h"
#ifdef YK_TEST_HOOKS
#include "../../include/yaik_hip_test.h"
#endif
#include "yk_device.h"

// ------------------------------------------------------------------------------------------------------------------
// a9: alpha tile-reject.  keep[mt] = 1 iff any of the 256 alphas of the aligned 16x16 block is non-zero (closed form of
// quadRecursion with maxMipLevel 3, EncoderContext.cpp:394-423); kept blocks grow the bounding box (boundingL/T/R/B, :416-422).
//
// ONE kernel (round 3: a clear of the flag map, a flagging kernel over single image rows with idempotent byte stores, and a
// bounding-box kernel over the flags -- three stream operations in every frame's chain of short kernels).  A work unit is
// (row of 16x16 tiles, 1024-pixel segment): the workgroup owns its 64 tiles, so it writes their flags outright (nothing to
// clear) and knows their box.  The plane is streamed with 16-byte loads, eight in flight per lane, a wave instruction covering
// 1 KB of one row; four adjacent lanes hold the 16 columns of one tile: 268 MB in 41 us = 6.5 TB/s without the box.
// The image-wide box WITHOUT atomics on its four words (every form of guarded atomicMin / atomicMax on them, dealt from the image's
// outside inwards or not, cost 23-85 us: a single address sustains ~88 atomics per microsecond, coherent guard reads are served by
// the same L2 channel one after the other): every unit leaves its box in a slot of its own (write-through store), arrivals are
// counted per group of 64 units and then per frame (<= 64 atomics per address), and the workgroup that arrives last folds the
// slots into bounds[8..11] = {min x0, min y0, max x1, max y1} ({9999999, 9999999, -1, -1} when nothing is kept).
// ------------------------------------------------------------------------------------------------------------------
typedef int yk_i4 __attribute__((ext_vector_type(4)));
#ifndef YK_ALPHA_INFLIGHT
#define YK_ALPHA_INFLIGHT 8                                                  // 16-byte loads a lane has in flight (two batches per unit)
#endif
#ifndef YK_ALPHA_THREADS
#define YK_ALPHA_THREADS 256                                                 // 256: a unit = 1024 pixels x 16 rows (four waves); 64: 256 pixels x 16 rows (one wave per workgroup)
#endif
#define YK_ALPHA_WAVES (YK_ALPHA_THREADS / 64)
#ifndef YK_ALPHA_ROWS
#define YK_ALPHA_ROWS 2                                                      // rows of 16x16 tiles per unit
#endif
__global__ __launch_bounds__(YK_ALPHA_THREADS) void yk_alpha_kernel(const int32_t* __restrict__ alpha0, int strideElems, int w, int h, int y0,
                                                       uint8_t* __restrict__ keep0, int mtW, int mtH, int32_t* __restrict__ bounds,
                                                       int nFrames, unsigned long long planeStride, unsigned long long keepStride,
                                                       int* __restrict__ unitBox0, uint32_t* __restrict__ arrive0) {
    __shared__ int s_box[YK_ALPHA_WAVES][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int vecPerRow = w >> 2;                                // int4 per image row (w is a multiple of 8)
    const int nSeg = (vecPerRow + YK_ALPHA_THREADS - 1) / YK_ALPHA_THREADS;   // YK_ALPHA_THREADS int4 per segment
    // a unit = YK_ALPHA_ROWS rows of tiles of one segment: the arrival protocol at its end (3-5 us in which the first wave keeps a slot of its SIMD)
    // is paid once per 2 x 16 rows, and the 2048 units of an 8192 x 8192 frame are exactly one round of resident workgroups
    const int nRowUnits = (mtH + YK_ALPHA_ROWS - 1) / YK_ALPHA_ROWS;
    const int nUnits = nSeg * nRowUnits, nGroups = (nUnits + 63) >> 6;
    for (long long uu = blockIdx.x; uu < (long long)nUnits * nFrames; uu += gridDim.x) {
        const int f = (int)(uu / nUnits), u = (int)(uu - (long long)f * nUnits);
        const int32_t* alpha = alpha0 + (size_t)f * planeStride;
        uint8_t* keep = keep0 + (size_t)f * keepStride;
        int* unitBox = unitBox0 + (size_t)f * (nUnits + nGroups) * 4;
        uint32_t* arrive = arrive0 + (size_t)f * (nGroups + 1);
        const int tyU = u / nSeg, seg = u - tyU * nSeg;
        const int xv = seg * YK_ALPHA_THREADS + (int)threadIdx.x;
        const bool inX = xv < vecPerRow;
        const bool tileLane = (lane & 3) == 0;
        const int tx = xv >> 2;
        int colLo = 0x7FFFFFFF, colHi = -1, rowLo = 0x7FFFFFFF, rowHi = -1;      // wave-uniform: kept tile columns / rows of this wave's share of the unit
#pragma unroll
        for (int r = 0; r < YK_ALPHA_ROWS; r++) {
            const int ty = tyU * YK_ALPHA_ROWS + r;
            if (ty >= mtH) break;
            const int32_t* col = alpha + (size_t)(ty * 16) * strideElems + (size_t)xv * 4;
            int nz = 0;
#pragma unroll
            for (int kb = 0; kb < 16 / YK_ALPHA_INFLIGHT; kb++) {
                yk_i4 a[YK_ALPHA_INFLIGHT];
#pragma unroll
                for (int k = 0; k < YK_ALPHA_INFLIGHT; k++) {
                    const int y = ty * 16 + kb * YK_ALPHA_INFLIGHT + k;
                    a[k] = (yk_i4){0, 0, 0, 0};
                    if (inX && y < h) a[k] = __builtin_nontemporal_load(reinterpret_cast<const yk_i4*>(col + (size_t)(kb * YK_ALPHA_INFLIGHT + k) * strideElems));
                }
#pragma unroll
                for (int k = 0; k < YK_ALPHA_INFLIGHT; k++) nz |= a[k].x | a[k].y | a[k].z | a[k].w;
            }
            const unsigned long long b = __ballot(nz != 0);
            const bool kept = ((b >> (lane & ~3)) & 0xFULL) != 0;
            if (tileLane && tx < mtW) keep[(size_t)ty * mtW + tx] = kept ? 1 : 0;
            // the kept tiles of this row (tile columns of this wave's 16 tiles)
            const unsigned long long kb64 = __ballot(tileLane && kept && tx < mtW);
            if (kb64) {
                const int t0 = seg * (YK_ALPHA_THREADS / 4) + wv * 16;
                colLo = min(colLo, t0 + ((__ffsll((long long)kb64) - 1) >> 2));
                colHi = max(colHi, t0 + ((63 - __clzll((long long)kb64)) >> 2));
                rowLo = min(rowLo, ty); rowHi = max(rowHi, ty);
            }
        }
        if (lane == 0) { s_box[wv][0] = colLo; s_box[wv][1] = colHi; s_box[wv][2] = rowLo; s_box[wv][3] = rowHi; }
        __syncthreads();
        // Only the first wave takes part in the arrival protocol (its round trips -- store acknowledged, then one or two returning atomics -- are
        // 4-5 us at the end of a 20 us workgroup); the other three go on (to their next unit, or out).  s_box is read before anything slow, and the
        // next iteration's barrier in front of its writes keeps the waves in step.
        if (wv == 0) {
            // unit boxes live in unitBox[0 .. nUnits), group boxes behind them; a box = two 64-bit words {x0 | y0 << 32, x1 | y1 << 32}, written
            // through to memory (the XCDs' L2s are not coherent with each other) and acknowledged before its owner counts as arrived
            unsigned long long* ub = reinterpret_cast<unsigned long long*>(unitBox);
            auto putBox = [&](const int slot, const int bx0, const int by0, const int bx1, const int by1) {
                __hip_atomic_store(&ub[slot * 2], ((unsigned long long)(uint32_t)by0 << 32) | (uint32_t)bx0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&ub[slot * 2 + 1], ((unsigned long long)(uint32_t)by1 << 32) | (uint32_t)bx1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_s_waitcnt(0);                                      // vmcnt(0): both stores have been acknowledged
            };
            auto foldBoxes = [&](const int first, const int n, int& x0, int& gy0, int& x1, int& gy1) {   // the whole wave: boxes [first, first + n)
                x0 = 9999999; gy0 = 9999999; x1 = -1; gy1 = -1;
                for (int k = lane; k < n; k += 64) {
                    const unsigned long long a = __hip_atomic_load(&ub[(first + k) * 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long bb = __hip_atomic_load(&ub[(first + k) * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    x0 = min(x0, (int)(uint32_t)a); gy0 = min(gy0, (int)(uint32_t)(a >> 32)); x1 = max(x1, (int)(uint32_t)bb); gy1 = max(gy1, (int)(uint32_t)(bb >> 32));
                }
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    x0 = min(x0, __shfl_xor(x0, d)); x1 = max(x1, __shfl_xor(x1, d));
                    gy0 = min(gy0, __shfl_xor(gy0, d)); gy1 = max(gy1, __shfl_xor(gy1, d));
                }
            };
            const int g = u >> 6, inGroup = min(64, nUnits - (g << 6));
            uint32_t lastOfGroup = 0;
            if (lane == 0) {
                int lo = s_box[0][0], hi = s_box[0][1], rlo = s_box[0][2], rhi = s_box[0][3];
#pragma unroll
                for (int k = 1; k < YK_ALPHA_WAVES; k++) { lo = min(lo, s_box[k][0]); hi = max(hi, s_box[k][1]); rlo = min(rlo, s_box[k][2]); rhi = max(rhi, s_box[k][3]); }
                const bool any = hi >= 0;
                putBox(u, any ? lo * 16 : 9999999, any ? y0 + rlo * 16 : 9999999, any ? hi * 16 + 16 : -1, any ? y0 + rhi * 16 + 16 : -1);
                lastOfGroup = __hip_atomic_fetch_add(&arrive[g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)inGroup - 1u ? 1u : 0u;
            }
            if (__builtin_amdgcn_readfirstlane((int)lastOfGroup)) {                  // the group's 64 boxes -> its group box (one round trip, anywhere in the launch)
                int x0, gy0, x1, gy1;
                foldBoxes(g << 6, inGroup, x0, gy0, x1, gy1);
                uint32_t lastOfFrame = 0;
                if (lane == 0) {
                    putBox(nUnits + g, x0, gy0, x1, gy1);
                    arrive[g] = 0u;                                                  // for the next frame (nobody else touches it any more)
                    lastOfFrame = __hip_atomic_fetch_add(&arrive[nGroups], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)nGroups - 1u ? 1u : 0u;
                }
                if (__builtin_amdgcn_readfirstlane((int)lastOfFrame)) {              // every unit of the frame has arrived: the group boxes -> the image's box
                    foldBoxes(nUnits, nGroups, x0, gy0, x1, gy1);
                    if (lane == 0) { int32_t* acc = bounds + (size_t)f * 16 + 8; acc[0] = x0; acc[1] = gy0; acc[2] = x1; acc[3] = gy1; arrive[nGroups] = 0u; }
                }
            }
        }
        if (uu + gridDim.x < (long long)nUnits * nFrames) __syncthreads();          // another unit follows: s_box is reused
    }
}

// test path only (the first-generation cross-check kernel reads the published form): bounds[0..3] = the box, bounds[4] = "bbox == whole image ->
// every reject discarded" (EncoderContext.cpp:1294, :1400-1403).  The library's own kernel derives the flag from the box.
__global__ void yk_alpha_publish_kernel(int32_t* __restrict__ bounds, int srcOff, int fullW, int fullH) {
    bounds += (size_t)blockIdx.x * 16;
    const int b0 = bounds[srcOff], b1 = bounds[srcOff + 1], b2 = bounds[srcOff + 2], b3 = bounds[srcOff + 3];
    bounds[0] = b0; bounds[1] = b1; bounds[2] = b2; bounds[3] = b3;
    bounds[4] = (b0 == 0 && b1 == 0 && b2 == fullW && b3 == fullH) ? 1 : 0;
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
// three planes (identical: the counts do not depend on the plane).  One workgroup of 1024 threads.
__device__ __forceinline__ void yk_scan2_body(uint32_t* __restrict__ blockCnt, uint32_t* __restrict__ blockSums, int nBlocks, uint32_t* __restrict__ totals, uint32_t* s_tmp) {
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
__global__ __launch_bounds__(1024) void yk_scan2_kernel(uint32_t* __restrict__ blockCnt, uint32_t* __restrict__ blockSums, int nBlocks,
                                                        uint32_t* __restrict__ totals, unsigned long long blockNStride) {
    __shared__ uint32_t s_tmp[32];
    yk_scan2_body(blockCnt + (size_t)blockIdx.x * blockNStride, blockSums + (size_t)blockIdx.x * blockNStride, nBlocks, totals + (size_t)blockIdx.x * 8, s_tmp);   // blockIdx.x = frame of a batch
}

// first level for the second-generation fused kernel, which neither adds to shared counters nor ORs into a shared map (atomics on one address are
// served one after the other and were 15-23 % of that kernel on noisy frames): every strip leaves the sums of its two runs of eight tiles in words
// of their own (runSums: nibbles / 16 | coded tiles << 16) and its four 16x16 bits in a byte of its own (bm0b).  Here the 128 run words of a scan
// block = 32 consecutive 16-byte pieces = one half-wave are added into the block's two counters (plain stores: nothing to clear, every word is
// rewritten every frame), and the four bytes of a 64x64 block are folded into its 16-bit word of the 16x16 map.
__global__ __launch_bounds__(1024) void yk_scan1r_kernel(const uint32_t* __restrict__ runSums, unsigned long long runStride, int nRuns,
                                                         uint32_t* __restrict__ blockCnt, unsigned long long blockNStride, int nBlocks,
                                                         const uint8_t* __restrict__ bm0b, unsigned long long bm0bStride, uint8_t* __restrict__ bitmap0,
                                                         unsigned long long bitmap0Stride, int nB64) {
    const size_t f = blockIdx.y;                                                  // frame of a batch
    const int g = (int)blockIdx.x * 1024 + (int)threadIdx.x;
    if (runSums) {
        const uint4* rs4 = reinterpret_cast<const uint4*>(runSums + f * runStride);
        const int n4 = (nRuns + 3) >> 2;                                          // the array is padded with zero words to a multiple of four
        const uint4 v = g < n4 ? rs4[g] : make_uint4(0u, 0u, 0u, 0u);
        uint32_t acc = v.x + v.y + v.z + v.w;                                     // both 16-bit halves at once: at most 128 x 32 and 128 x 8 per block
        acc += __shfl_xor(acc, 16, 32); acc += __shfl_xor(acc, 8, 32); acc += __shfl_xor(acc, 4, 32); acc += __shfl_xor(acc, 2, 32); acc += __shfl_xor(acc, 1, 32);
        const int blk = g >> 5;
        if ((threadIdx.x & 31) == 0 && blk < nBlocks) {
            uint2* dst = reinterpret_cast<uint2*>(blockCnt + f * blockNStride) + blk;
            *dst = make_uint2(16u * (acc & 0xFFFFu), acc >> 16);
        }
    }
    if (g < nB64) {
        const uint32_t b = reinterpret_cast<const uint32_t*>(bm0b + f * bm0bStride)[g];   // the four strips' bytes of block g, 4 bits each
        uint16_t* dst = reinterpret_cast<uint16_t*>(bitmap0 + f * bitmap0Stride);
        dst[g] = (uint16_t)((b & 0xFu) | ((b >> 4) & 0xF0u) | ((b >> 8) & 0xF00u) | ((b >> 12) & 0xF000u));
        if (g == nB64 - 1 && (nB64 & 1)) dst[nB64] = 0;                           // the padding half of the last 32-bit word (word-wise consumers)
    }
}

// One workgroup packs the nibbles of 1024 consecutive tiles, the three planes one after the other: a tile's count is the same in the three
// planes, so ONE scan serves them all (round 3 ran a workgroup and two block scans per plane).  Every tile holds a multiple of 16 nibbles
// (16 per uncovered 4x4 quadrant), so every stream offset is a multiple of 8 bytes: after the scan, four lanes per tile copy 8-byte pieces
// of the tile's slot straight to its place in the stream, reading only the bytes that exist.
__global__ __launch_bounds__(1024) void yk_pack_kernel(const uint8_t* __restrict__ tileCount, const uint16_t* __restrict__ tileDef, const uint2* __restrict__ tileInfo,
                                                       const uint8_t* __restrict__ slots, size_t T8, const uint32_t* __restrict__ blockSums, int nBlocks,
                                                       uint16_t* __restrict__ defsOut, uint32_t* __restrict__ nibOut, size_t nibStrideWords, YkFrameStrides fs,
                                                       const uint32_t* __restrict__ blockCnt, uint32_t* __restrict__ totals) {
    __shared__ uint32_t s_tmp[32];
    __shared__ unsigned long long s_pre[17];
    __shared__ uint32_t s_off[YK_SCAN_TILE];
    __shared__ uint8_t s_cnt[YK_SCAN_TILE];
    {   // blockIdx.z = frame of a batch
        const size_t f = blockIdx.z;
        tileCount += f * fs.tileCount; tileDef += f * fs.tileDef; slots += f * fs.slots; blockSums += f * fs.blockN;
        if (tileInfo) tileInfo += f * fs.tileInfo;
        defsOut += f * fs.defsOut; nibOut += f * (fs.nibOut / 4);
        if (blockCnt) { blockCnt += f * fs.blockN; totals += f * 8; }
    }
    // Second scan level inside this kernel (second-generation fused kernel: the per-block sums come from yk_scan1r_kernel): the workgroup adds the
    // sums of the blocks in front of it (at most 8 KB from L2) instead of waiting for a one-workgroup kernel in the frame's chain of launches.
    uint32_t preN = 0, preD = 0;
    if (blockCnt) {
        unsigned long long acc = 0;                                               // coded tiles << 32 | nibbles
        for (int k = threadIdx.x; k < (int)blockIdx.x; k += 1024) { const uint2 v = reinterpret_cast<const uint2*>(blockCnt)[k]; acc += ((unsigned long long)v.y << 32) | v.x; }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) acc += __shfl_xor(acc, d);
        if ((threadIdx.x & 63) == 0) s_pre[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) { unsigned long long t = 0; for (int k = 0; k < 16; k++) t += s_pre[k]; s_pre[16] = t; }
        __syncthreads();
        preN = (uint32_t)s_pre[16]; preD = (uint32_t)(s_pre[16] >> 32);
        if (blockIdx.x == (unsigned)nBlocks - 1 && threadIdx.x < 3) {            // the totals of the three planes (identical)
            const uint2 v = reinterpret_cast<const uint2*>(blockCnt)[nBlocks - 1];
            totals[threadIdx.x * 2] = preD + v.y; totals[threadIdx.x * 2 + 1] = preN + v.x;
        }
    }
    const size_t i0 = (size_t)blockIdx.x * YK_SCAN_TILE;
    {
        const size_t i = i0 + threadIdx.x;
        // second-generation fused kernel: one record per tile {def0 | def1 << 16, def2 | count << 16}; first generation: count and definition arrays
        uint32_t c = 0, d01 = 0, d2 = 0;
        if (i < T8) {
            if (tileInfo) { const uint2 ti = tileInfo[i]; c = ti.y >> 16; d01 = ti.x; d2 = ti.y & 0xFFFFu; }
            else { c = tileCount[i]; if (c) { d01 = (uint32_t)tileDef[i] | ((uint32_t)tileDef[T8 + i] << 16); d2 = tileDef[2 * T8 + i]; } }
        }
        uint32_t totN, totD;
        const uint32_t en = yk_block_exscan(c, s_tmp, &totN);
        const uint32_t ed = yk_block_exscan(c ? 1u : 0u, s_tmp, &totD);
        const uint32_t baseN = blockCnt ? preN : blockSums[(size_t)blockIdx.x * 2], baseD = blockCnt ? preD : blockSums[(size_t)blockIdx.x * 2 + 1];     // same for the three planes
        s_off[threadIdx.x] = (baseN + en) >> 1;                                   // byte offset inside a plane's stream
        s_cnt[threadIdx.x] = (uint8_t)c;
        if (c) {
            defsOut[baseD + ed] = (uint16_t)d01; defsOut[T8 + baseD + ed] = (uint16_t)(d01 >> 16); defsOut[2 * T8 + baseD + ed] = (uint16_t)d2;
        }
    }
    __syncthreads();
    const int piece = threadIdx.x & 3;
#pragma unroll
    for (int p = 0; p < 3; p++) {
        uint8_t* out = reinterpret_cast<uint8_t*>(nibOut + (size_t)p * nibStrideWords);
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int t = it * 256 + (threadIdx.x >> 2);
            const size_t i = i0 + t;
            if (i < T8 && piece * 8 < (s_cnt[t] >> 1))
                *reinterpret_cast<uint2*>(out + s_off[t] + piece * 8) = *reinterpret_cast<const uint2*>(slots + ((size_t)p * T8 + i) * YK_SLOT + piece * 8);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------------------
// test hook: the launcher of an independent second implementation of the fused kernel (same YkEncodeParams, same outputs)
static int (*g_crossCheckLauncher)(hipStream_t, const YkEncodeParams*) = nullptr;
#ifdef YK_TEST_HOOKS
extern "C" int yk_set_cross_check_launcher(void* fn) { g_crossCheckLauncher = reinterpret_cast<int (*)(hipStream_t, const YkEncodeParams*)>(fn); return YK_OK; }
#endif

int yk_launch_alpha(yk_ctx* c, bool batch) {
    const int F = batch ? c->nFrames : 1;
    uint8_t* keep = batch ? c->B.keep : c->keep;
    int32_t* bounds = batch ? c->B.bounds : c->bounds;
    const int32_t* alpha = batch ? c->B.plane[3] : c->plane[3];
    const int nSeg = (c->fullW / 4 + YK_ALPHA_THREADS - 1) / YK_ALPHA_THREADS;
    const long long nUnits = (long long)nSeg * ((c->mtH + YK_ALPHA_ROWS - 1) / YK_ALPHA_ROWS) * F;
    const long long maxWG = 4096LL * (256 / YK_ALPHA_THREADS);
    hipLaunchKernelGGL(yk_alpha_kernel, dim3((unsigned)(nUnits < maxWG ? nUnits : maxWG)), dim3(YK_ALPHA_THREADS), 0, c->stream, alpha, c->strideElems, c->fullW, c->h, c->y0,
                       keep, c->mtW, c->mtH, bounds, F, (unsigned long long)c->fs.plane, (unsigned long long)c->fs.keep, c->alphaUnitBox, c->alphaArrive);
    YK_HIP(c, hipGetLastError());
    c->boundsOff = 8;                                           // whole image / batch: the accumulators are the box; a stripe caller replaces it (yk_alpha_finish)
    return YK_OK;
}

int yk_launch_alpha_finish(yk_ctx* c, const int32_t* globalBBox) {
    // whole image: the box is where yk_alpha_kernel accumulated it; stripes: the host-combined box goes to bounds[0..4]
    if (globalBBox) {
        int32_t b[5] = { globalBBox[0], globalBBox[1], globalBBox[2], globalBBox[3], 0 };
        b[4] = (b[0] == 0 && b[1] == 0 && b[2] == c->fullW && b[3] == c->fullH) ? 1 : 0;
        YK_HIP(c, hipMemcpyAsync(c->bounds, b, sizeof b, hipMemcpyHostToDevice, c->stream));
        c->boundsOff = 0;
    }
    return YK_OK;
}

int yk_launch_encode(yk_ctx* c, int rejectFactor, int mode3BitOnly, int wantDst, bool batch) {
    YkEncodeParams P;
    for (int i = 0; i < 4; i++) P.plane[i] = batch ? c->B.plane[i] : c->plane[i];
    P.strideElems = c->strideElems; P.w = c->fullW; P.h = c->h; P.hAvail = c->h + c->halo; P.y0 = c->y0; P.fullH = c->fullH;
    P.rejectFactor = rejectFactor; P.startMode = mode3BitOnly ? 3 : 0; P.wantDst = wantDst; P.ablate = c->ablate;
    P.keep = (c->nPlanes == 4) ? (batch ? c->B.keep : c->keep) : nullptr;
    P.bounds = (c->nPlanes == 4) ? (batch ? c->B.bounds : c->bounds) + c->boundsOff : nullptr;   // {x0, y0, x1, y1}; the kernel derives the discard rule
    for (int i = 0; i < 7; i++) P.bitmap[i] = batch ? c->B.bitmap[i] : c->bitmap[i];
    P.coverage = batch ? c->B.coverage : c->coverage; P.tileDef = batch ? c->B.tileDef : c->tileDef;
    P.tileCount = batch ? c->B.tileCount : c->tileCount; P.slots = batch ? c->B.slots : c->slots;
    P.blockCnt = c->kernelVersion == 2 ? (batch ? c->B.blockCnt : c->blockCnt) : nullptr;
    for (int i = 0; i < 3; i++) P.dst[i] = c->dst[i];
    P.tilesW = c->tilesW; P.tilesH = c->tilesH; P.mtW = c->mtW; P.mtH = c->mtH;
    P.xBB64 = (c->fullW + 63) / 64; P.yBB64 = (c->h + 63) / 64; P.xBB32 = (c->fullW + 31) / 32; P.yBB32 = (c->h + 31) / 32;
    P.nFrames = batch ? c->nFrames : 1; P.fs = c->fs;
    P.qtab = c->qtab;
    P.small = c->B.small;                                       // frame f of an array sits at its offset + f * stride (yk_alloc_image)
    for (int i = 0; i < 7; i++) P.oBm[i] = (uint32_t)(c->B.bitmap[i] - c->B.small);
    P.oBm0b = (uint32_t)(c->B.bm0b - c->B.small); P.oCov = (uint32_t)(reinterpret_cast<uint8_t*>(c->B.coverage) - c->B.small);
    P.oInfo = (uint32_t)(reinterpret_cast<uint8_t*>(c->B.tileInfo) - c->B.small); P.oRun = (uint32_t)(reinterpret_cast<uint8_t*>(c->B.runSums) - c->B.small);
    if (!batch && c->curFrame) {                                // a selected frame of a batch encoded on its own
        for (int i = 0; i < 7; i++) P.oBm[i] += (uint32_t)(c->curFrame * c->fs.bitmap[i]);
        P.oBm0b += (uint32_t)(c->curFrame * c->fs.bm0b); P.oCov += (uint32_t)(c->curFrame * c->fs.coverage * 2);
        P.oInfo += (uint32_t)(c->curFrame * c->fs.tileInfo * 8); P.oRun += (uint32_t)(c->curFrame * c->fs.runSums * 4);
    }
    P.pixCache = nullptr; c->pixCacheValid = false;
    if (c->pixCacheOn && !batch && c->kernelVersion == 2 && c->nFrames == 1) {
        if (!c->pixCache) YK_HIP(c, hipMalloc(&c->pixCache, (size_t)P.xBB64 * 64 * ((size_t)(c->h + 15) / 16 * 16) * 4 + 4096));
        P.pixCache = c->pixCache; c->pixCacheValid = true;
    }
    if (c->kernelVersion == 2) return yk_launch_encode2(c, P);
    // version 1 = the cross-check implementation of the test suite (tests/csrc/yk_encode_v1.hip), registered at run time
    if (batch) return yk_fail(c, YK_ERR_STATE, "batches need kernel version 2");
    if (c->nPlanes == 4) {                                      // it reads the published form bounds[0..4]
        hipLaunchKernelGGL(yk_alpha_publish_kernel, dim3(1), dim3(1), 0, c->stream, c->bounds, c->boundsOff, c->fullW, c->fullH);
        P.bounds = c->bounds;
    }
    if (!g_crossCheckLauncher) return yk_fail(c, YK_ERR_STATE, "kernel version 1 is not part of this library: register it with yk_set_cross_check_launcher");
    if (g_crossCheckLauncher(c->stream, &P) != 0) return yk_fail(c, YK_ERR_HIP, "cross-check kernel launch", hipGetLastError());
    return YK_OK;
}

int yk_launch_pack(yk_ctx* c, bool batch) {
    const size_t T8 = (size_t)c->tilesW * c->tilesH;
    const int nb = c->nScanBlocks, F = batch ? c->nFrames : 1;
    uint32_t* blockCnt = batch ? c->B.blockCnt : c->blockCnt; uint32_t* blockSums = batch ? c->B.blockSums : c->blockSums;
    uint32_t* totals = batch ? c->B.totals : c->totals; uint8_t* nibOut = batch ? c->B.nibOut : c->nibOut;
    if (c->kernelVersion != 2) hipLaunchKernelGGL(yk_scan1_kernel, dim3(nb), dim3(1024), 0, c->stream, c->tileCount, T8, c->blockCnt);
    // second-generation fused kernel: per-run sums and per-strip 16x16 bytes instead of atomics, per-tile records instead of count / definition arrays
    const bool v2 = c->kernelVersion == 2;
    const size_t f0 = batch ? 0 : (size_t)c->curFrame;
    const uint32_t* runSums = (v2 && (c->tilesW & 7) == 0) ? c->B.runSums + f0 * c->fs.runSums : nullptr;
    const uint8_t* bm0b = v2 ? c->B.bm0b + f0 * c->fs.bm0b : nullptr;
    const uint2* tileInfo = v2 ? c->B.tileInfo + f0 * c->fs.tileInfo : nullptr;
    const int nB64 = ((c->fullW + 63) / 64) * ((c->h + 63) / 64);
    if (v2) {
        const int nRuns = (int)((T8 + 7) / 8), n4 = (nRuns + 3) >> 2;
        const int items = runSums ? (n4 > nB64 ? n4 : nB64) : nB64;
        hipLaunchKernelGGL(yk_scan1r_kernel, dim3((unsigned)((items + 1023) / 1024), F), dim3(1024), 0, c->stream, runSums, (unsigned long long)c->fs.runSums, nRuns,
                           blockCnt, (unsigned long long)c->fs.blockN, nb, bm0b, (unsigned long long)c->fs.bm0b,
                           batch ? c->B.bitmap[0] : c->bitmap[0], (unsigned long long)c->fs.bitmap[0], nB64);
    }
    if (!runSums) hipLaunchKernelGGL(yk_scan2_kernel, dim3(F), dim3(1024), 0, c->stream, blockCnt, blockSums, nb, totals, (unsigned long long)c->fs.blockN);
    hipLaunchKernelGGL(yk_pack_kernel, dim3(nb, 1, F), dim3(1024), 0, c->stream, batch ? c->B.tileCount : c->tileCount, batch ? c->B.tileDef : c->tileDef, tileInfo,
                       batch ? c->B.slots : c->slots, T8, blockSums, nb, batch ? c->B.defsOut : c->defsOut, reinterpret_cast<uint32_t*>(nibOut), c->nibStride / 4, c->fs,
                       runSums ? blockCnt : (const uint32_t*)nullptr, totals);
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

// yk_r1_magic against the C expression of GetValueModel1 for every delta, minCol and pixel value the 1-D path can meet
__global__ void yk_selftest_r1magic_kernel(int* mismatches) {
    const int delta = blockIdx.x, x = threadIdx.x;               // delta 0..255, x = v - minCol 0..delta
    if (x > delta) return;
    for (int minCol = 0; minCol + delta <= 255; minCol++) {
        uint32_t A, B;
        yk_r1_magic(delta, minCol, &A, &B);
        const int v = minCol + x;
        int idx = 0;
        if (delta) { const int n = (v - minCol) * 15 + (delta >> 1) - 1; idx = n < 0 ? -1 : n / delta; }
        const uint32_t got = (__umul24((uint32_t)v, A) + B) >> 20;
        if (got != (uint32_t)(1 + idx)) atomicAdd(mismatches, 1);
    }
}

#ifdef YK_TEST_HOOKS
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
    else if (which == 4) hipLaunchKernelGGL(yk_selftest_r1magic_kernel, dim3(256), dim3(256), 0, c->stream, d);
    else { (void)hipFree(d); return yk_fail(c, YK_ERR_BAD_ARG, "unknown selftest"); }
    YK_HIP(c, hipMemcpyAsync(result, d, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    YK_HIP(c, hipStreamSynchronize(c->stream));
    (void)hipFree(d);
    return YK_OK;
}
#endif
