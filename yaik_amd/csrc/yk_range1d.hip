// yk_range1d.hip — a15: the live 1-D range path, EncoderContext::DynamicTileCompressor (encoder/EncoderContext.cpp:8398-8522)
// with FindAndRemoveMostUsedColor (:8335), Model1 (:8358), GetValueModel1 (:8383).  This is the range coder Convert() actually
// runs (:9451-9465) and the one Decompress1D decodes ('1DTL' chunk), so it closes the wire-level round trip.
//
// Per 8x8 tile and plane, over the pixels of the 4x4 quadrants no gradient tile covers (`map` pixel at the quadrant origin == 0,
// :8420-8431):  color0 = right-most mode of the 256-bin histogram of CompressF(v,255) (= v), clamped to 1..254; bins color0+-1
// removed; (minCol, delta) = extent of what is left; one byte per pixel: 0 if |v-color0| <= 1 else 1 + ((v-minCol)*15 + (delta>>1) - 1)/delta.
// One wave per tile, one pixel per lane; the per-tile results land in fixed slots and are compacted into the reference's
// streams (pixel bytes of plane 0,1,2 appended; 3 parameter bytes per tile likewise) by a prefix sum over the tile grid.
#include "yk_common.h"
#include "yk_device.h"

__global__ __launch_bounds__(256) void yk_range1d_kernel(const int32_t* __restrict__ pR, const int32_t* __restrict__ pG, const int32_t* __restrict__ pB,
                                                         int strideElems, int w, int h, const uint16_t* __restrict__ coverage, int mtW,
                                                         int tilesW, size_t T8, uint8_t* __restrict__ slots, uint8_t* __restrict__ params,
                                                         uint32_t* __restrict__ cntTiles, uint32_t* __restrict__ cntPix) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t ti = (size_t)blockIdx.x * 4 + wave;
    if (ti >= T8) return;
    const int tx = (int)(ti % tilesW), ty = (int)(ti / tilesW);
    const int x = lane & 7, y = lane >> 3;
    const int gx = tx * 8 + x, gy = ty * 8 + y;
    // quadrant coverage from the macro-tile coverage word (bit = cellY*4 + cellX)
    const uint32_t cw = coverage[(size_t)(gy >> 4) * mtW + (gx >> 4)];
    const int cellX0 = ((tx * 8) >> 2) & 3, cellY0 = ((ty * 8) >> 2) & 3;
    const uint32_t cwT = coverage[(size_t)((ty * 8) >> 4) * mtW + ((tx * 8) >> 4)];
    const bool u00 = !((cwT >> (cellY0 * 4 + cellX0)) & 1), u10 = !((cwT >> (cellY0 * 4 + cellX0 + 1)) & 1);
    const bool u01 = !((cwT >> ((cellY0 + 1) * 4 + cellX0)) & 1), u11 = !((cwT >> ((cellY0 + 1) * 4 + cellX0 + 1)) & 1);
    (void)cw;
    const bool valid = (y < 4) ? ((x < 4) ? u00 : u10) : ((x < 4) ? u01 : u11);
    const int nTop = (int)u00 + (int)u10, nBot = (int)u01 + (int)u11;
    const int nPix = 16 * (nTop + nBot);
    if (lane == 0) { cntTiles[ti] = nPix ? 1u : 0u; cntPix[ti] = (uint32_t)nPix; }
    if (nPix == 0) return;                                     // wave-uniform
    // position among the tile's emitted pixels: top half rows (left then right quadrant), then bottom half (:8420-8453)
    const int pos = (y < 4) ? (y * 4 * nTop + ((x >= 4) ? 4 * (int)u00 : 0) + (x & 3))
                            : (16 * nTop + (y - 4) * 4 * nBot + ((x >= 4) ? 4 * (int)u01 : 0) + (x & 3));
    const int32_t* planes[3] = { pR, pG, pB };
    for (int p = 0; p < 3; p++) {
        const int v = valid ? (planes[p][(size_t)gy * strideElems + gx] & 255) : -1;      // CompressF(v,255) == v
        // histogram mode: count of equal valid values, right-most maximum wins (:8339-8344)
        int cnt = 0;
        for (int j = 0; j < 64; j++) {
            const int vj = __shfl(v, j);
            cnt += (vj == v) ? 1 : 0;
        }
        int key = valid ? ((cnt << 8) | v) : -1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) key = max(key, __shfl_xor(key, d));
        int color0 = key & 255;
        if (color0 == 0) color0 = 1;
        if (color0 == 255) color0 = 254;
        const bool isC0 = valid && v >= color0 - 1 && v <= color0 + 1;
        // Model1 over the remaining histogram (:8358-8381)
        int mn = (valid && !isC0) ? v : 99999, mx = (valid && !isC0) ? v : -99999;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { mn = min(mn, __shfl_xor(mn, d)); mx = max(mx, __shfl_xor(mx, d)); }
        int minCol = 0, delta = 0;
        if (mn != 99999) { minCol = mn; delta = mx - mn; }
        if (valid) {
            int out = 0;
            if (!isC0) {
                const int idx = delta ? (((v - minCol) * 15) + ((delta >> 1) - 1)) / delta : 0;     // GetValueModel1 (:8383-8391)
                out = 1 + idx;
            }
            slots[((size_t)p * T8 + ti) * 64 + pos] = (uint8_t)out;
        }
        if (lane == 0) {
            uint8_t* q = params + ((size_t)p * T8 + ti) * 4;
            q[0] = (uint8_t)color0; q[1] = (uint8_t)minCol; q[2] = (uint8_t)delta;
        }
    }
}

__global__ __launch_bounds__(1024) void yk_range1d_pack_kernel(const uint32_t* __restrict__ cntTiles, const uint32_t* __restrict__ cntPix,
                                                               const uint32_t* __restrict__ baseTiles, const uint32_t* __restrict__ basePix,
                                                               const uint32_t* __restrict__ totals, size_t T8, const uint8_t* __restrict__ slots,
                                                               const uint8_t* __restrict__ params, uint8_t* __restrict__ pixOut, uint8_t* __restrict__ typeOut) {
    __shared__ uint32_t s_tmp[32];
    const int p = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    uint32_t tot;
    const uint32_t et = yk_block_exscan(i < T8 ? cntTiles[i] : 0u, s_tmp, &tot);
    const uint32_t ep = yk_block_exscan(i < T8 ? cntPix[i] : 0u, s_tmp, &tot);
    if (i >= T8 || !cntTiles[i]) return;
    const size_t to = ((size_t)p * totals[0] + baseTiles[blockIdx.x] + et) * 3, po = (size_t)p * totals[1] + basePix[blockIdx.x] + ep;
    const uint8_t* q = params + ((size_t)p * T8 + i) * 4;
    typeOut[to] = q[0]; typeOut[to + 1] = q[1]; typeOut[to + 2] = q[2];
    const uint8_t* s = slots + ((size_t)p * T8 + i) * 64;
    const uint32_t n = cntPix[i];
    for (uint32_t k = 0; k < n; k++) pixOut[po + k] = s[k];
}

extern "C" {

int yk_range1d_encode(yk_ctx* c) {
    if (!c) return YK_ERR_BAD_ARG;
    if (!c->encoded) return yk_fail(c, YK_ERR_STATE, "yk_encode_tiles first (the 1-D path codes what the gradient passes left uncovered)");
    YK_HIP(c, hipSetDevice(c->device));
    const size_t T8 = (size_t)c->tilesW * c->tilesH, nb = (T8 + 1023) / 1024;
    if (!c->r1Slots) {
        YK_HIP(c, hipMalloc(&c->r1Slots, 3 * T8 * 64));
        YK_HIP(c, hipMalloc(&c->r1Params, 3 * T8 * 4));
        YK_HIP(c, hipMalloc(&c->r1Cnt, (2 * T8 + 2 * nb + 64) * sizeof(uint32_t)));
        YK_HIP(c, hipMalloc(&c->r1Pix, 3 * T8 * 64 + 64));
        YK_HIP(c, hipMalloc(&c->r1Type, 3 * T8 * 3 + 64));
    }
    uint32_t* cT = c->r1Cnt; uint32_t* cP = cT + T8; uint32_t* bT = cP + T8; uint32_t* bP = bT + nb + 16; uint32_t* tot = bP + nb + 16;
    hipLaunchKernelGGL(yk_range1d_kernel, dim3((unsigned)((T8 + 3) / 4)), dim3(256), 0, c->stream, c->plane[0], c->plane[1], c->plane[2], c->strideElems,
                       c->fullW, c->h, c->coverage, c->mtW, c->tilesW, T8, c->r1Slots, c->r1Params, cT, cP);
    hipLaunchKernelGGL(yk_u32_blocksum_kernel, dim3((unsigned)nb), dim3(1024), 0, c->stream, cT, T8, bT);
    hipLaunchKernelGGL(yk_u32_scanblocks_kernel, dim3(1), dim3(1024), 0, c->stream, bT, (int)nb, tot);
    hipLaunchKernelGGL(yk_u32_blocksum_kernel, dim3((unsigned)nb), dim3(1024), 0, c->stream, cP, T8, bP);
    hipLaunchKernelGGL(yk_u32_scanblocks_kernel, dim3(1), dim3(1024), 0, c->stream, bP, (int)nb, tot + 1);
    hipLaunchKernelGGL(yk_range1d_pack_kernel, dim3((unsigned)nb, 3), dim3(1024), 0, c->stream, cT, cP, bT, bP, tot, T8, c->r1Slots, c->r1Params, c->r1Pix, c->r1Type);
    YK_HIP(c, hipGetLastError());
    uint32_t t[2];
    YK_HIP(c, hipMemcpyAsync(t, tot, sizeof t, hipMemcpyDeviceToHost, c->stream));
    YK_HIP(c, hipStreamSynchronize(c->stream));
    c->r1Tiles = t[0]; c->r1PixCount = t[1]; c->r1Ready = true;
    return YK_OK;
}

int yk_range1d_streams(yk_ctx* c, uint8_t* hostPix, size_t capPix, size_t* nPix, uint8_t* hostType, size_t capType, size_t* nType) {
    if (!c) return YK_ERR_BAD_ARG;
    if (!c->r1Ready) return yk_fail(c, YK_ERR_STATE, "yk_range1d_encode first");
    const size_t np = (size_t)c->r1PixCount * 3, nt = (size_t)c->r1Tiles * 9;
    if (nPix) *nPix = np;
    if (nType) *nType = nt;
    YK_HIP(c, hipSetDevice(c->device));
    if (hostPix) { if (capPix < np) return yk_fail(c, YK_ERR_RANGE, "pixel stream buffer too small"); if (np) YK_HIP(c, hipMemcpyAsync(hostPix, c->r1Pix, np, hipMemcpyDeviceToHost, c->stream)); }
    if (hostType) { if (capType < nt) return yk_fail(c, YK_ERR_RANGE, "type stream buffer too small"); if (nt) YK_HIP(c, hipMemcpyAsync(hostType, c->r1Type, nt, hipMemcpyDeviceToHost, c->stream)); }
    YK_HIP(c, hipStreamSynchronize(c->stream));
    return YK_OK;
}

}  // extern "C"
