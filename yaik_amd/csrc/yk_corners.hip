// yk_corners.hip — the corner-colour streams of the gradient passes (`rgbStream`, encoder/EncoderContext.cpp:3783-3785,
// emission :4113-4132, cross-pass de-duplication through `mappedRGB` :4001-4021).
//
// The reference emits, for every accepted tile in scan order, the corners TL,TR,BL,BR whose lattice point has not been
// emitted by any earlier tile of this or an earlier pass.  That is a first-toucher problem:
//   1. owner[lattice point] = min over all accepted tiles touching it of key = pass<<27 | bitIndex<<2 | corner
//      (bitIndex = the tile's position in the swizzled bitmap = the reference's scan order `pos`)        -- atomicMin
//   2. per pass: every bitmap word counts the corners its tiles own, an exclusive scan gives the byte offsets,
//      and the owners write CompressF(Round6(v),250) for R,G,B.
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

__global__ __launch_bounds__(256) void yk_corner_owner_kernel(const uint32_t* __restrict__ bitmap, size_t nWords, int pass, int w, int latW,
                                                              uint32_t* __restrict__ owner) {
    const size_t wi = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= nWords) return;
    uint32_t bits = bitmap[wi];
    const PassGeo g = yk_pass_geo(pass, w);
    while (bits) {
        const int b = __ffs(bits) - 1; bits &= bits - 1;
        const uint32_t pos = (uint32_t)(wi * 32 + b);
        int x, y; yk_tile_from_bit(g, pos, x, y);
        const int lx = x >> 2, ly = y >> 2, dx = 1 << (g.sx - 2), dy = 1 << (g.sy - 2);
        const uint32_t key = ((uint32_t)pass << 27) | (pos << 2);
        atomicMin(&owner[(size_t)ly * latW + lx], key | 0u);
        atomicMin(&owner[(size_t)ly * latW + lx + dx], key | 1u);
        atomicMin(&owner[(size_t)(ly + dy) * latW + lx], key | 2u);
        atomicMin(&owner[(size_t)(ly + dy) * latW + lx + dx], key | 3u);
    }
}

// One thread per tile slot (bit) of the pass's bitmap, 1024 slots per workgroup.  COUNT: corners owned per workgroup.
// EMIT: exclusive scan inside the workgroup + the scanned workgroup bases = byte offset of every owned corner.
template <bool EMIT>
__global__ __launch_bounds__(1024) void yk_corner_stream_kernel(const uint32_t* __restrict__ bitmap, size_t nBits, int pass, int w, int latW,
                                                                const uint32_t* __restrict__ owner, uint32_t* __restrict__ blockSums,
                                                                const int32_t* const __restrict__ pR, const int32_t* const __restrict__ pG,
                                                                const int32_t* const __restrict__ pB, int strideElems,
                                                                uint8_t* __restrict__ out, uint32_t* __restrict__ edgeIdx, int latH, int hAvail) {
    __shared__ uint32_t s_tmp[32];
    const size_t pos = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const bool set = pos < nBits && ((bitmap[pos >> 5] >> (pos & 31)) & 1u);
    const PassGeo g = yk_pass_geo(pass, w);
    const int dx = 1 << (g.sx - 2), dy = 1 << (g.sy - 2);
    int x = 0, y = 0;
    uint32_t own = 0, cnt = 0;
    const uint32_t key = ((uint32_t)pass << 27) | ((uint32_t)pos << 2);
    if (set) {
        yk_tile_from_bit(g, (uint32_t)pos, x, y);
        const size_t l0 = (size_t)(y >> 2) * latW + (x >> 2);
        own = (owner[l0] == (key | 0u) ? 1u : 0u) | (owner[l0 + dx] == (key | 1u) ? 2u : 0u) |
              (owner[l0 + (size_t)dy * latW] == (key | 2u) ? 4u : 0u) | (owner[l0 + (size_t)dy * latW + dx] == (key | 3u) ? 8u : 0u);
        cnt = (uint32_t)__popc(own);
    }
    uint32_t tot;
    const uint32_t ex = yk_block_exscan(cnt, s_tmp, &tot);
    if (!EMIT) { if (threadIdx.x == 0) blockSums[blockIdx.x] = tot; return; }
    uint32_t off = (blockSums[blockIdx.x] + ex) * 3u;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (!((own >> k) & 1u)) continue;
        const int lx = (x >> 2) + ((k & 1) ? dx : 0), ly = (y >> 2) + ((k & 2) ? dy : 0);
        // GetPixelValue clamp (:3853-3856); a stripe's bottom lattice row is its halo row (= the next stripe's first row)
        const size_t src = (size_t)min(ly * 4, hAvail - 1) * strideElems + min(lx * 4, w - 1);
        // stripes: where along this pass's stream the first and last lattice rows were emitted (root-side de-duplication)
        if (ly == 0) edgeIdx[lx] = off / 3u;
        if (ly == latH - 1) edgeIdx[latW + lx] = off / 3u;
        const int v[3] = { pR[src], pG[src], pB[src] };
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const int r6 = (v[ch] & ~3) | (v[ch] >> 6);                                       // Round6 (:3183)
            out[off + ch] = (uint8_t)((r6 * 250 + 127) / 255);                                 // CompressF(.,colorCompressionQuad=250) (:3191)
        }
        off += 3;
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
    size_t nbTot = 0; size_t nbOf[7], bitsOf[7];
    for (int p = 0; p < 7; p++) { bitsOf[p] = c->bitmapBytes[p] * 8; nbOf[p] = (bitsOf[p] + 1023) / 1024; nbTot += nbOf[p]; }
    if (!c->cornerScratch) { c->cornerScratchElems = nbTot + 64; YK_HIP(c, hipMalloc(&c->cornerScratch, c->cornerScratchElems * 4)); }
    uint32_t* totalDev = c->cornerScratch + nbTot;
    if (!c->cornerEdgeIdx) YK_HIP(c, hipMalloc(&c->cornerEdgeIdx, (size_t)latW * 2 * 4));
    { int rc = yk_stage_begin(c, YK_STAGE_CORNERS); if (rc) return rc; }
    YK_HIP(c, hipMemsetAsync(c->cornerEdgeIdx, 0xFF, (size_t)latW * 2 * 4, c->stream));
    YK_HIP(c, hipMemsetAsync(c->latticeOwner, 0xFF, lat * 4, c->stream));
    for (int p = 0; p < 7; p++) {
        const size_t nWords = (c->bitmapBytes[p] + 3) / 4;      // bitmap allocations are padded by 16 bytes; pass 0 words may be half used
        if (c->bitmapBytes[p] & 3) YK_HIP(c, hipMemsetAsync(c->bitmap[p] + c->bitmapBytes[p], 0, 4 - (c->bitmapBytes[p] & 3), c->stream));
        hipLaunchKernelGGL(yk_corner_owner_kernel, dim3((unsigned)((nWords + 255) / 256)), dim3(256), 0, c->stream,
                           reinterpret_cast<const uint32_t*>(c->bitmap[p]), nWords, p, w, latW, c->latticeOwner);
    }
    uint32_t* blockSums = c->cornerScratch;
    for (int p = 0; p < 7; p++) {
        const unsigned nb = (unsigned)nbOf[p];
        const uint32_t* bm = reinterpret_cast<const uint32_t*>(c->bitmap[p]);
        hipLaunchKernelGGL(yk_corner_stream_kernel<false>, dim3(nb), dim3(1024), 0, c->stream, bm, bitsOf[p], p, w, latW, c->latticeOwner, blockSums,
                           c->plane[0], c->plane[1], c->plane[2], c->strideElems, (uint8_t*)nullptr, (uint32_t*)nullptr, latH, c->h + c->halo);
        hipLaunchKernelGGL(yk_u32_scanblocks_kernel, dim3(1), dim3(1024), 0, c->stream, blockSums, (int)nb, totalDev + p);
        hipLaunchKernelGGL(yk_corner_stream_kernel<true>, dim3(nb), dim3(1024), 0, c->stream, bm, bitsOf[p], p, w, latW, c->latticeOwner, blockSums,
                           c->plane[0], c->plane[1], c->plane[2], c->strideElems, c->cornerStream + region * p, c->cornerEdgeIdx, latH, c->h + c->halo);
        blockSums += nb;
    }
    YK_HIP(c, hipGetLastError());
    { int rc = yk_stage_end(c, YK_STAGE_CORNERS); if (rc) return rc; }
    uint32_t totals[7];
    YK_HIP(c, hipMemcpyAsync(totals, totalDev, sizeof totals, hipMemcpyDeviceToHost, c->stream));
    YK_HIP(c, hipStreamSynchronize(c->stream));
    for (int p = 0; p < 7; p++) { c->cornerOff[p] = region * p; c->cornerBytes[p] = (size_t)totals[p] * 3; }
    c->cornersReady = true;
    return YK_OK;
}

extern "C" int yk_gradient_corners(yk_ctx* c, int pass, uint8_t* hostOut, size_t cap, size_t* nBytes) {
    if (!c || pass < 0 || pass >= 7) return YK_ERR_BAD_ARG;
    if (!c->encoded) return yk_fail(c, YK_ERR_STATE, "yk_encode_tiles first");
    YK_HIP(c, hipSetDevice(c->device));
    if (!c->cornersReady) { int rc = yk_launch_corners(c); if (rc) return rc; }
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
