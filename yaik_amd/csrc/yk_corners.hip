// yk_corners.hip — the corner-colour streams of the gradient passes (`rgbStream`, encoder/EncoderContext.cpp:3783-3785,
// emission :4113-4132, cross-pass de-duplication through `mappedRGB` :4001-4021).
//
// The reference emits, for every accepted tile in scan order, the corners TL,TR,BL,BR whose lattice point has not been
// emitted by any earlier tile of this or an earlier pass.  That is a first-toucher problem:
//   1. owner[lattice point] = min over all accepted tiles touching it of key = pass<<27 | bitIndex<<2 | corner
//      (bitIndex = the tile's position in the swizzled bitmap = the reference's scan order `pos`)        -- atomicMin
//   2. every block of 1024 tile slots counts the corners its tiles own, ONE exclusive scan over the blocks of all passes gives the
//      byte offsets (relative to the pass's first block), and the owners write CompressF(Round6(v),250) for R,G,B.
// All bitmaps stay on the device; only the finished streams are copied out.
#include "yk_common.h"
#include "yk_device.h"

struct PassGeo { int sx, sy, bigX, bigY, bitCount, xBB, tilesPerRow; };

__host__ __device__ static inline PassGeo yk_pass_geo(int pass, int w) {
    const int sxs[7] = { 4, 4, 3, 3, 3, 2, 2 }, sys[7] = { 4, 3, 4, 3, 2, 3, 2 };
    PassGeo g; g.sx = sxs[pass]; g.sy = sys[pass];
    g.bigX = g.sx == 2 ? 32 : 64; g.bigY = g.sy == 2 ? 32 : 64;        // getSwizzleSize, include/YAIK_private.h:212-276
    g.tilesPerRow = g.bigX >> g.sx;
    g.bitCount = g.tilesPerRow * (g.bigY >> g.sy);
    g.xBB = (w + g.bigX - 1) / g.bigX;
    return g;
}

__device__ __forceinline__ void yk_tile_from_bit(const PassGeo& g, uint32_t pos, int& x, int& y) {
    const uint32_t blk = pos / g.bitCount, t = pos % g.bitCount;
    x = (int)(blk % g.xBB) * g.bigX + (int)(t % g.tilesPerRow) * (1 << g.sx);
    y = (int)(blk / g.xBB) * g.bigY + (int)(t / g.tilesPerRow) * (1 << g.sy);
}

// The seven passes share every launch: bitmap words (owner) and 1024-slot blocks (count / emit) of all passes are laid end to end.
struct CornerPlan {
    uint32_t wordStart[8];          // first bitmap word of pass p in the concatenated word space ([7] = total)
    uint32_t blockStart[8];         // first 1024-slot block of pass p ([7] = total)
    unsigned long long bits[7];     // tile slots of pass p
    const uint32_t* bm[7];
};
__device__ __forceinline__ int yk_plan_find(const uint32_t (&start)[8], uint32_t i) {
    int p = 0;
#pragma unroll
    for (int k = 1; k < 7; k++) p += (i >= start[k]) ? 1 : 0;
    return p;
}

__global__ __launch_bounds__(256) void yk_corner_owner_kernel(const CornerPlan pl, int w, int latW, uint32_t* __restrict__ owner) {
    // One thread per bitmap byte (8 slots of one swizzle block: the block's coordinates are computed once, see yk_corner_stream_kernel).
    // Measured alternatives on the 8192x8192 bench frame (131 k accepted 16x16 tiles): one thread per word 41 us, per byte 41 us, per byte
    // with the atomics of corners a lower-positioned neighbour tile of the same pass also touches left out 47 us, per slot with that
    // filter 51 us (16 us of it just starting 10.7 M threads).  What does pay is the filter WITHOUT any look-up: among the 8 tiles of the
    // thread's own byte (15 atomics instead of 32 for a dense 2 x 4 group of 16x16 tiles).
    const uint32_t gi = blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= pl.wordStart[7] * 4u) return;
    const int pass = yk_plan_find(pl.wordStart, gi >> 2);
    const uint32_t bi = gi - pl.wordStart[pass] * 4u;
    const uint32_t byte = reinterpret_cast<const uint8_t*>(pl.bm[pass])[bi];
    if (!byte) return;
    const PassGeo g = yk_pass_geo(pass, w);
    const uint32_t pos0 = bi * 8u, blk = pos0 / (uint32_t)g.bitCount, t0 = pos0 % (uint32_t)g.bitCount;
    const int bx0 = (int)(blk % (uint32_t)g.xBB) * g.bigX, by0 = (int)(blk / (uint32_t)g.xBB) * g.bigY;
    const int tprShift = __ffs(g.tilesPerRow) - 1, dx = 1 << (g.sx - 2), dy = 1 << (g.sy - 2);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (!((byte >> k) & 1u)) continue;
        const uint32_t t = t0 + (uint32_t)k;
        const int x = bx0 + (int)((t & (uint32_t)(g.tilesPerRow - 1)) << g.sx), y = by0 + (int)((t >> tprShift) << g.sy);
        const int lx = x >> 2, ly = y >> 2;
        const uint32_t key = ((uint32_t)pass << 27) | ((pos0 + (uint32_t)k) << 2);
        // a corner an EARLIER tile of this same byte also touches belongs to that tile (smaller scan position): no atomic for it.  The byte
        // is one row of 8 tiles, or two rows of 4 when the swizzle block is 4 tiles wide: left, upper, upper-left, upper-right neighbours.
        const int col = (int)(t & (uint32_t)(g.tilesPerRow - 1));
        const bool two = g.tilesPerRow == 4;
        const bool left = k >= 1 && col != 0 && ((byte >> (k - 1)) & 1u);
        const bool up = two && k >= 4 && ((byte >> (k - 4)) & 1u);
        const bool upLeft = two && k >= 5 && col != 0 && ((byte >> (k - 5)) & 1u);
        const bool upRight = two && k >= 4 && col != 3 && ((byte >> (k - 3)) & 1u);
        if (!(left || up || upLeft)) atomicMin(&owner[(size_t)ly * latW + lx], key | 0u);
        if (!(up || upRight)) atomicMin(&owner[(size_t)ly * latW + lx + dx], key | 1u);
        if (!left) atomicMin(&owner[(size_t)(ly + dy) * latW + lx], key | 2u);
        atomicMin(&owner[(size_t)(ly + dy) * latW + lx + dx], key | 3u);
    }
}

// One thread per bitmap BYTE (8 tile slots), 1024 bytes per workgroup.  (Round 1 ran one thread per slot: a block scan per 1024 slots of
// mostly empty maps cost more than the work it ordered; one thread per 32-bit word serialises 32 dependent gather rounds in the dense
// 16x16 regions: 149 us for the emit launch.)  The 32 owner look-ups of a byte are issued together, unconditionally (slots that are not
// set read entry 0 and are masked), so a thread waits for memory once.  COUNT: corners owned per thread (kept as bytes for the emit
// launch) and per workgroup.  EMIT: exclusive scan inside the workgroup + the scanned workgroup bases (relative to the pass's first
// block) = byte offset of the thread's first owned corner; a thread walks its tiles in bit order = the reference's scan order.
template <bool EMIT>
__global__ __launch_bounds__(1024) void yk_corner_stream_kernel(const CornerPlan pl, int w, int latW,
                                                                const uint32_t* __restrict__ owner, uint32_t* __restrict__ blockSums, uint32_t* __restrict__ perThread,
                                                                const int32_t* const __restrict__ pR, const int32_t* const __restrict__ pG,
                                                                const int32_t* const __restrict__ pB, int strideElems,
                                                                uint8_t* __restrict__ out0, size_t region, uint32_t* __restrict__ edgeIdx, int latH, int hAvail) {
    __shared__ uint32_t s_tmp[32];
    const int pass = yk_plan_find(pl.blockStart, blockIdx.x);
    const uint32_t bi = (blockIdx.x - pl.blockStart[pass]) * 1024u + threadIdx.x;           // byte of the pass's bitmap
    const uint32_t nBytes = (pl.wordStart[pass + 1] - pl.wordStart[pass]) * 4u;
    const size_t ti = (size_t)pl.wordStart[pass] * 4u + bi;                                   // thread index over all passes
    const uint32_t byte = bi < nBytes ? reinterpret_cast<const uint8_t*>(pl.bm[pass])[bi] : 0u;
    const PassGeo g = yk_pass_geo(pass, w);
    const int dx = 1 << (g.sx - 2), dy = 1 << (g.sy - 2);
    // COUNT leaves the thread's ownership bits (4 per tile slot) for EMIT, which then neither repeats the 32 owner look-ups nor waits for them
    uint32_t cnt = 0, ownBits = 0;
    if (EMIT) { ownBits = bi < nBytes ? perThread[ti] : 0u; cnt = (uint32_t)__popc(ownBits); }
    uint32_t own[8];
    int tx[8], ty[8];
    if (!EMIT || cnt) {
        // the 8 slots of a byte lie in one swizzle block (every block holds a multiple of 8 tiles); tiles per row / per block are powers of two
        const uint32_t pos0 = bi * 8u, blk = pos0 / (uint32_t)g.bitCount, t0 = pos0 % (uint32_t)g.bitCount;
        const int bx0 = (int)(blk % (uint32_t)g.xBB) * g.bigX, by0 = (int)(blk / (uint32_t)g.xBB) * g.bigY;
        const int tprShift = __ffs(g.tilesPerRow) - 1;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t t = t0 + (uint32_t)k;
            tx[k] = bx0 + (int)((t & (uint32_t)(g.tilesPerRow - 1)) << g.sx); ty[k] = by0 + (int)((t >> tprShift) << g.sy);
        }
        if (EMIT) {
#pragma unroll
            for (int k = 0; k < 8; k++) own[k] = (ownBits >> (4 * k)) & 15u;
        } else {
            uint32_t o[8][4];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const bool set = (byte >> k) & 1u;
                const size_t l0 = set ? (size_t)(ty[k] >> 2) * latW + (tx[k] >> 2) : 0;
                const size_t ddx = set ? dx : 0, ddy = set ? (size_t)dy * latW : 0;
                o[k][0] = owner[l0]; o[k][1] = owner[l0 + ddx]; o[k][2] = owner[l0 + ddy]; o[k][3] = owner[l0 + ddy + ddx];
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t key = ((uint32_t)pass << 27) | ((bi * 8u + (uint32_t)k) << 2);
                const bool set = (byte >> k) & 1u;
                own[k] = set ? ((o[k][0] == (key | 0u) ? 1u : 0u) | (o[k][1] == (key | 1u) ? 2u : 0u) | (o[k][2] == (key | 2u) ? 4u : 0u) | (o[k][3] == (key | 3u) ? 8u : 0u)) : 0u;
            }
        }
    }
    if (!EMIT) {
#pragma unroll
        for (int k = 0; k < 8; k++) { cnt += (uint32_t)__popc(own[k]); ownBits |= own[k] << (4 * k); }
        uint32_t tot;
        yk_block_exscan(cnt, s_tmp, &tot);
        if (bi < nBytes) perThread[ti] = ownBits;
        if (threadIdx.x == 0) blockSums[blockIdx.x] = tot;
        return;
    }
    uint32_t tot;
    const uint32_t ex = yk_block_exscan(cnt, s_tmp, &tot);
    // The workgroup's colours are one contiguous run of the pass's stream: they are collected in LDS and leave as 4-byte stores (three
    // single-byte stores per colour from scattered lanes were most of this kernel's time); a workgroup with more colours than the LDS
    // image holds (possible only with many small tiles owning all four corners) stores directly.
    constexpr uint32_t kCap = 8192;                                          // colours
    __shared__ uint32_t s_out[kCap * 3 / 4];
    __shared__ uint32_t s_li[kCap];                                          // lattice index of the workgroup's j-th colour, in stream order
    const uint32_t blockOff = (blockSums[blockIdx.x] - blockSums[pl.blockStart[pass]]) * 3u;
    uint8_t* __restrict__ out = out0 + region * pass;
    const bool viaLds = tot <= kCap;
    uint8_t* const s_bytes = reinterpret_cast<uint8_t*>(s_out);
    auto colour = [&](const uint32_t li, const uint32_t j, const bool toLds) {   // j-th colour of the workgroup: lattice point li
        const int ly = (int)(li / (uint32_t)latW), lx = (int)(li - (uint32_t)ly * (uint32_t)latW);
        // GetPixelValue clamp (:3853-3856); a stripe's bottom lattice row is its halo row (= the next stripe's first row)
        const size_t src = (size_t)min(ly * 4, hAvail - 1) * strideElems + min(lx * 4, w - 1);
        // stripes: where along this pass's stream the first and last lattice rows were emitted (root-side de-duplication)
        if (ly == 0) edgeIdx[lx] = blockOff / 3u + j;
        if (ly == latH - 1) edgeIdx[latW + lx] = blockOff / 3u + j;
        const int v[3] = { pR[src], pG[src], pB[src] };
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const int r6 = (v[ch] & ~3) | (v[ch] >> 6);                                           // Round6 (:3183)
            const uint8_t q = (uint8_t)((r6 * 250 + 127) / 255);                                   // CompressF(.,colorCompressionQuad=250) (:3191)
            if (toLds) s_bytes[j * 3u + ch] = q; else out[blockOff + j * 3u + ch] = q;
        }
    };
    // phase 1: every thread lists the lattice points it owns at its place in the run (scan order); phase 2: one thread per colour, so the
    // three sample loads of the workgroup's colours are all in flight together instead of one owned corner after the other per thread
    if (cnt) {
        uint32_t j = ex;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (!own[k]) continue;
#pragma unroll
            for (int c4 = 0; c4 < 4; c4++) {
                if (!((own[k] >> c4) & 1u)) continue;
                const uint32_t li = (uint32_t)((ty[k] >> 2) + ((c4 & 2) ? dy : 0)) * (uint32_t)latW + (uint32_t)((tx[k] >> 2) + ((c4 & 1) ? dx : 0));
                if (viaLds) s_li[j] = li; else colour(li, j, false);
                j++;
            }
        }
    }
    if (!viaLds) return;
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < tot; j += 1024) colour(s_li[j], j, true);
    __syncthreads();
    // copy out: unaligned head bytes, whole words, tail bytes (the run starts at an arbitrary byte of the stream)
    const uint32_t outBytes = tot * 3u;
    uint8_t* const dst = out + blockOff;
    const uint32_t head = min(outBytes, (uint32_t)((4u - (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 3u)) & 3u));
    if (threadIdx.x < head) dst[threadIdx.x] = s_bytes[threadIdx.x];
    const uint32_t nWords = (outBytes - head) >> 2;
    for (uint32_t i = threadIdx.x; i < nWords; i += 1024) {
        const uint32_t b = head + i * 4u;                                    // LDS side is unaligned by `head`: assemble from bytes
        const uint32_t wv = (uint32_t)s_bytes[b] | ((uint32_t)s_bytes[b + 1] << 8) | ((uint32_t)s_bytes[b + 2] << 16) | ((uint32_t)s_bytes[b + 3] << 24);
        *reinterpret_cast<uint32_t*>(dst + b) = wv;
    }
    const uint32_t tail0 = head + nWords * 4u;
    if (threadIdx.x < outBytes - tail0) dst[tail0 + threadIdx.x] = s_bytes[tail0 + threadIdx.x];
}

// exclusive prefix of all block sums in place (one workgroup: thread t owns a run of consecutive blocks) + the corners per pass
__global__ __launch_bounds__(1024) void yk_corner_scan_kernel(uint32_t* __restrict__ blockSums, const CornerPlan pl, uint32_t* __restrict__ passTotals) {
    __shared__ uint32_t s_tmp[32];
    __shared__ uint32_t s_total;
    const uint32_t n = pl.blockStart[7], per = (n + 1023) / 1024;
    const uint32_t a = min(threadIdx.x * per, n), b = min(a + per, n);
    uint32_t sum = 0;
    for (uint32_t i = a; i < b; i++) sum += blockSums[i];
    uint32_t tot;
    uint32_t run = yk_block_exscan(sum, s_tmp, &tot);
    for (uint32_t i = a; i < b; i++) { const uint32_t v = blockSums[i]; blockSums[i] = run; run += v; }
    if (threadIdx.x == 0) s_total = tot;
    __syncthreads();
    if (threadIdx.x < 7) {
        const uint32_t lo = blockSums[pl.blockStart[threadIdx.x]];
        const uint32_t hi = pl.blockStart[threadIdx.x + 1] < n ? blockSums[pl.blockStart[threadIdx.x + 1]] : s_total;
        passTotals[threadIdx.x] = (pl.blockStart[threadIdx.x] < n) ? hi - lo : 0u;
    }
}

int yk_launch_corners(yk_ctx* c) {
    // Row stripes: the handle de-duplicates inside its own rows; its first and last lattice rows are shared with the
    // neighbouring stripes and are reconciled on the root from yk_gradient_corner_edges (see distributed.merge_corner_streams).
    const int w = c->fullW, h = c->h;
    const int latW = w / 4 + 1, latH = h / 4 + 1;
    const size_t lat = (size_t)latW * latH;
    if (!c->latticeOwner) { YK_HIP(c, hipMalloc(&c->latticeOwner, lat * 4)); c->latticeElems = lat; }
    // every lattice point is emitted at most once over the seven passes, so one buffer of lat*3 bytes holds all streams back to
    // back; the passes are laid out at their worst-case offsets (pass p after everything passes < p could emit) = 7 regions
    const size_t region = lat * 3 + 16;
    if (!c->cornerStream) { c->cornerCap = region * 7; YK_HIP(c, hipMalloc(&c->cornerStream, c->cornerCap)); }
    CornerPlan pl;
    pl.wordStart[0] = 0; pl.blockStart[0] = 0;
    for (int p = 0; p < 7; p++) {
        pl.bits[p] = (unsigned long long)c->bitmapBytes[p] * 8;
        pl.bm[p] = reinterpret_cast<const uint32_t*>(c->bitmap[p]);
        pl.wordStart[p + 1] = pl.wordStart[p] + (uint32_t)((c->bitmapBytes[p] + 3) / 4);   // bitmap allocations are padded by 16 bytes; pass 0 words may be half used
        pl.blockStart[p + 1] = pl.blockStart[p] + ((pl.wordStart[p + 1] - pl.wordStart[p]) * 4u + 1023u) / 1024u;   // 1024 bitmap bytes per workgroup
    }
    const size_t nbTot = pl.blockStart[7], nWordsTot = pl.wordStart[7];
    // scratch: [block sums | 7 totals (+ pad) | ownership bits per thread (one word per bitmap byte)]
    if (!c->cornerScratch) { c->cornerScratchElems = nbTot + 64 + nWordsTot * 4; YK_HIP(c, hipMalloc(&c->cornerScratch, c->cornerScratchElems * 4)); }
    uint32_t* blockSums = c->cornerScratch;
    uint32_t* totalDev = c->cornerScratch + nbTot;
    uint32_t* perThread = c->cornerScratch + nbTot + 64;
    if (!c->cornerEdgeIdx) YK_HIP(c, hipMalloc(&c->cornerEdgeIdx, (size_t)latW * 2 * 4));
    { int rc = yk_stage_begin(c, YK_STAGE_CORNERS); if (rc) return rc; }
    YK_HIP(c, hipMemsetAsync(c->cornerEdgeIdx, 0xFF, (size_t)latW * 2 * 4, c->stream));
    YK_HIP(c, hipMemsetAsync(c->latticeOwner, 0xFF, lat * 4, c->stream));
    for (int p = 0; p < 7; p++)
        if (c->bitmapBytes[p] & 3) YK_HIP(c, hipMemsetAsync(c->bitmap[p] + c->bitmapBytes[p], 0, 4 - (c->bitmapBytes[p] & 3), c->stream));
    // 2 clears + 4 launches for the seven passes (round 1: 2 + 28): owner of every lattice point, corners per block, one scan, emission
    hipLaunchKernelGGL(yk_corner_owner_kernel, dim3((pl.wordStart[7] * 4u + 255u) / 256u), dim3(256), 0, c->stream, pl, w, latW, c->latticeOwner);
    hipLaunchKernelGGL(yk_corner_stream_kernel<false>, dim3((unsigned)nbTot), dim3(1024), 0, c->stream, pl, w, latW, c->latticeOwner, blockSums, perThread,
                       c->plane[0], c->plane[1], c->plane[2], c->strideElems, (uint8_t*)nullptr, region, (uint32_t*)nullptr, latH, c->h + c->halo);
    hipLaunchKernelGGL(yk_corner_scan_kernel, dim3(1), dim3(1024), 0, c->stream, blockSums, pl, totalDev);
    hipLaunchKernelGGL(yk_corner_stream_kernel<true>, dim3((unsigned)nbTot), dim3(1024), 0, c->stream, pl, w, latW, c->latticeOwner, blockSums, perThread,
                       c->plane[0], c->plane[1], c->plane[2], c->strideElems, c->cornerStream, region, c->cornerEdgeIdx, latH, c->h + c->halo);
    YK_HIP(c, hipGetLastError());
    { int rc = yk_stage_end(c, YK_STAGE_CORNERS); if (rc) return rc; }
    // the stream lengths stay on the device until somebody asks for a stream (yk_corners_finish): a caller that keeps frames in flight
    // (bench.py --stage all, the C++ mirror's pipelined conversion) is not stopped here
    for (int p = 0; p < 7; p++) c->cornerOff[p] = region * p;
    c->cornerTotalsDev = totalDev; c->cornerTotalsPending = true;
    c->cornersReady = true;
    return YK_OK;
}

int yk_corners_finish(yk_ctx* c) {
    if (!c->cornerTotalsPending) return YK_OK;
    uint32_t totals[7];
    YK_HIP(c, hipMemcpyAsync(totals, c->cornerTotalsDev, sizeof totals, hipMemcpyDeviceToHost, c->stream));
    YK_HIP(c, hipStreamSynchronize(c->stream));
    for (int p = 0; p < 7; p++) c->cornerBytes[p] = (size_t)totals[p] * 3;
    c->cornerTotalsPending = false;
    return YK_OK;
}

extern "C" int yk_gradient_corners(yk_ctx* c, int pass, uint8_t* hostOut, size_t cap, size_t* nBytes) {
    if (!c || pass < 0 || pass >= 7) return YK_ERR_BAD_ARG;
    if (!c->encoded) return yk_fail(c, YK_ERR_STATE, "yk_encode_tiles first");
    YK_HIP(c, hipSetDevice(c->device));
    if (!c->cornersReady) { int rc = yk_launch_corners(c); if (rc) return rc; }
    { int rc = yk_corners_finish(c); if (rc) return rc; }
    if (nBytes) *nBytes = c->cornerBytes[pass];
    if (hostOut) {
        if (cap < c->cornerBytes[pass]) return yk_fail(c, YK_ERR_RANGE, "corner buffer too small");
        if (c->cornerBytes[pass]) {
            YK_HIP(c, hipMemcpyAsync(hostOut, c->cornerStream + c->cornerOff[pass], c->cornerBytes[pass], hipMemcpyDeviceToHost, c->stream));
            YK_HIP(c, hipStreamSynchronize(c->stream));
        }
    }
    return YK_OK;
}

extern "C" int yk_gradient_corners_device(yk_ctx* c, int pass, const uint8_t** dev, size_t* nBytes) {
    if (!c || pass < 0 || pass >= 7 || !dev || !nBytes) return YK_ERR_BAD_ARG;
    if (!c->encoded) return yk_fail(c, YK_ERR_STATE, "yk_encode_tiles first");
    YK_HIP(c, hipSetDevice(c->device));
    if (!c->cornersReady) { int rc = yk_launch_corners(c); if (rc) return rc; }
    { int rc = yk_corners_finish(c); if (rc) return rc; }
    *dev = c->cornerStream + c->cornerOff[pass]; *nBytes = c->cornerBytes[pass];
    return YK_OK;
}

extern "C" int yk_gradient_corners_run(yk_ctx* c) {
    if (!c) return YK_ERR_BAD_ARG;
    if (!c->encoded) return yk_fail(c, YK_ERR_STATE, "yk_encode_tiles first");
    YK_HIP(c, hipSetDevice(c->device));
    return yk_launch_corners(c);
}

extern "C" int yk_gradient_corner_edges(yk_ctx* c, uint32_t* hostKeys, uint32_t* hostIndex, size_t capElems) {
    if (!c || !hostKeys || !hostIndex) return YK_ERR_BAD_ARG;
    if (!c->encoded) return yk_fail(c, YK_ERR_STATE, "yk_encode_tiles first");
    YK_HIP(c, hipSetDevice(c->device));
    if (!c->cornersReady) { int rc = yk_launch_corners(c); if (rc) return rc; }
    const size_t latW = (size_t)c->fullW / 4 + 1, latH = (size_t)c->h / 4 + 1;
    if (capElems < 2 * latW) return yk_fail(c, YK_ERR_RANGE, "edge buffers need 2 * (w/4 + 1) elements");
    YK_HIP(c, hipMemcpyAsync(hostKeys, c->latticeOwner, latW * 4, hipMemcpyDeviceToHost, c->stream));
    YK_HIP(c, hipMemcpyAsync(hostKeys + latW, c->latticeOwner + (latH - 1) * latW, latW * 4, hipMemcpyDeviceToHost, c->stream));
    YK_HIP(c, hipMemcpyAsync(hostIndex, c->cornerEdgeIdx, 2 * latW * 4, hipMemcpyDeviceToHost, c->stream));
    YK_HIP(c, hipStreamSynchronize(c->stream));
    return YK_OK;
}
