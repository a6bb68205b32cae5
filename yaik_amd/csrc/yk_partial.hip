// yk_partial.hip — EncoderContext::FittingQuadSmooth with nullable planes (encoder/EncoderContext.cpp:3710-4363): one more gradient
// pass over a SUBSET of the planes (PlaneBit :3715), run after the seven RGB passes of the fused kernel.  The reference lists six such
// 4x4 passes in Convert() (RB, RG, GB, R, G, B; :9261-9415) and keeps them switched off (`if (0)`, `#if 0`); the function itself takes
// any tile shape, and so does this file.  These passes are not on the timed path: the kernels are plain, one thread per tile slot.
//
// What changes against an RGB pass:
//   * absent planes read as 0 everywhere (:3857-3862, :3910-3912): they never reject;
//   * a tile is allowed when its top-left pixel is uncovered in every PRESENT plane (:3871-3875): coverage is kept per plane
//     (mapSmoothTile[p]; smoothMap / mipmapMask are painted for all, :4029-4037), here as one bit per 4x4 cell and plane;
//   * a corner emits one byte per present plane that has not seen that lattice point (mappedRGB[p], :4001-4021, :4113-4132).
// Tiles of one pass never overlap, so all decisions of a pass are independent given the state before it; the corner stream is the
// same first-toucher problem as in yk_corners.hip (owner by scan position, counts, scan, emit).
#include "yk_common.h"
#include "yk_device.h"

struct PPGeo { int sx, sy, bigX, bigY, bitCount, xBB, tilesPerRow; };
__host__ __device__ static inline PPGeo yk_pp_geo(int sx, int sy, int w) {
    PPGeo g; g.sx = sx; g.sy = sy;
    g.bigX = sx == 2 ? 32 : 64; g.bigY = sy == 2 ? 32 : 64;             // getSwizzleSize, include/YAIK_private.h:212-276
    g.tilesPerRow = g.bigX >> sx; g.bitCount = g.tilesPerRow * (g.bigY >> sy);
    g.xBB = (w + g.bigX - 1) / g.bigX;
    return g;
}
__device__ __forceinline__ void yk_pp_tile(const PPGeo& g, uint32_t pos, int& x, int& y) {
    const uint32_t blk = pos / g.bitCount, t = pos % g.bitCount;
    x = (int)(blk % g.xBB) * g.bigX + (int)(t % g.tilesPerRow) * (1 << g.sx);
    y = (int)(blk / g.xBB) * g.bigY + (int)(t / g.tilesPerRow) * (1 << g.sy);
}
__device__ __forceinline__ int yk_pp_r6(int v) { return (v & ~3) | (v >> 6); }                       // Round6  :3183
__device__ __forceinline__ int yk_pp_r6p(int v) { v = min(v + 1, 255); return (v & ~3) | (v >> 6); } // Round6P :3202

struct PPPlanes { const int32_t* p[3]; int strideElems, w, h, hAvail; };
__device__ __forceinline__ int yk_pp_px(const PPPlanes& P, int n, int x, int y) {                     // Plane::GetPixelValue clamp (framework.h:116-121)
    return P.p[n][(size_t)min(y, P.hAvail - 1) * P.strideElems + min(x, P.w - 1)];
}

// first use after an encode: every plane starts from the common coverage, every lattice point an RGB pass emitted is known to all planes
__global__ void yk_pp_init_kernel(const uint16_t* __restrict__ coverage, size_t nMT, uint16_t* __restrict__ covCh, size_t covStride,
                                  const uint32_t* __restrict__ latticeOwner, size_t lat, uint8_t* __restrict__ mapped3) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nMT) { const uint16_t v = coverage[i]; covCh[i] = v; covCh[covStride + i] = v; covCh[2 * covStride + i] = v; }
    if (i < lat) mapped3[i] = latticeOwner[i] != 0xFFFFFFFFu ? 7 : 0;
}

// decision: one thread per tile slot of the pass's swizzled bitmap (the six variants of :3929-3991, integer arithmetic as in the reference)
__global__ __launch_bounds__(256) void yk_pp_decide_kernel(PPPlanes P, PPGeo g, size_t nBits, int planeBit, int rf, const uint16_t* __restrict__ covCh, size_t covStride,
                                                           int mtW, uint32_t* __restrict__ bitmap, uint32_t* __restrict__ accepted) {
    const size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= nBits) return;
    int x, y; yk_pp_tile(g, (uint32_t)pos, x, y);
    const int TX = 1 << g.sx, TY = 1 << g.sy;
    if (x + TX > P.w || y + TY > P.h) return;                                // partial tiles are skipped (:3818, :3826)
    const size_t mt = (size_t)(y >> 4) * mtW + (x >> 4);
    const int cbit = ((y >> 2) & 3) * 4 + ((x >> 2) & 3);
    for (int n = 0; n < 3; n++) if (((planeBit >> n) & 1) && ((covCh[n * covStride + mt] >> cbit) & 1)) return;      // :3871-3875
    int c[3][4], c6[3][4], cp[3][4];
    for (int n = 0; n < 3; n++) for (int k = 0; k < 4; k++) {
        const int v = ((planeBit >> n) & 1) ? yk_pp_px(P, n, x + ((k & 1) ? TX : 0), y + ((k & 2) ? TY : 0)) : 0;
        c[n][k] = v; c6[n][k] = yk_pp_r6(v); cp[n][k] = yk_pp_r6p(v);
    }
    bool rej[6] = { false, false, false, false, false, false };
    const int rounding = (1 << 19) - 1;
    for (int dy = 0; dy < TY; dy++) {
        const int tF = 1024 - dy * (1024 >> g.sy), bF = 1024 - tF;           // weight4 / 8 / 16 (:3735-3737)
        for (int dx = 0; dx < TX; dx++) {
            const int lF = 1024 - dx * (1024 >> g.sx), rF = 1024 - lF;
            for (int n = 0; n < 3; n++) {
                if (!((planeBit >> n) & 1)) continue;                        // an absent plane reads 0 against a blend of 0: never rejects
                const int cur = P.p[n][(size_t)(y + dy) * P.strideElems + (x + dx)];
                const int S = (c[n][0] * lF + c[n][1] * rF) * tF + (c[n][2] * lF + c[n][3] * rF) * bF;
                const int S6 = (c6[n][0] * lF + c6[n][1] * rF) * tF + (c6[n][2] * lF + c6[n][3] * rF) * bF;
                const int SP = (cp[n][0] * lF + cp[n][1] * rF) * tF + (cp[n][2] * lF + cp[n][3] * rF) * bF;
                rej[0] |= abs(cur - ((S + rounding) >> 20)) > rf;  rej[2] |= abs(cur - (S >> 20)) > rf;
                rej[1] |= abs(cur - ((S6 + rounding) >> 20)) > rf; rej[3] |= abs(cur - (S6 >> 20)) > rf;
                rej[5] |= abs(cur - ((SP + rounding) >> 20)) > rf; rej[4] |= abs(cur - (SP >> 20)) > rf;
            }
        }
    }
    if (rej[0] && rej[1] && rej[2] && rej[3] && rej[4] && rej[5]) return;    // :3998
    atomicOr(&bitmap[pos >> 5], 1u << (pos & 31));
    atomicAdd(accepted, 1u);
}

// paint (:4029-4037): after the decisions (a tile must not see the coverage of its own pass: tiles of a pass are disjoint, so it could not
// matter, but the launch boundary keeps that obvious): per present plane + the common coverage (smoothMap)
__global__ __launch_bounds__(256) void yk_pp_paint_kernel(PPGeo g, size_t nBits, int planeBit, const uint32_t* __restrict__ bitmap, uint32_t* __restrict__ covCh32,
                                                          size_t covStride, uint32_t* __restrict__ coverage32, int mtW) {
    const size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= nBits || !((bitmap[pos >> 5] >> (pos & 31)) & 1u)) return;
    int x, y; yk_pp_tile(g, (uint32_t)pos, x, y);
    for (int cy = y >> 2; cy < (y >> 2) + (1 << (g.sy - 2)); cy++) for (int cx = x >> 2; cx < (x >> 2) + (1 << (g.sx - 2)); cx++) {
        const size_t mt = (size_t)(cy >> 2) * mtW + (cx >> 2);
        const uint32_t bit = 1u << (((cy & 3) * 4 + (cx & 3)) + 16 * (mt & 1));
        atomicOr(&coverage32[mt >> 1], bit);
        for (int n = 0; n < 3; n++) if ((planeBit >> n) & 1) atomicOr(&covCh32[(n * covStride + mt) >> 1], 1u << (((cy & 3) * 4 + (cx & 3)) + 16 * ((n * covStride + mt) & 1)));
    }
}

__global__ __launch_bounds__(256) void yk_pp_owner_kernel(PPGeo g, size_t nBits, int planeBit, const uint32_t* __restrict__ bitmap, int latW,
                                                          const uint8_t* __restrict__ mapped3, uint32_t* __restrict__ owner) {
    const size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= nBits || !((bitmap[pos >> 5] >> (pos & 31)) & 1u)) return;
    int x, y; yk_pp_tile(g, (uint32_t)pos, x, y);
    const int dx = 1 << (g.sx - 2), dy = 1 << (g.sy - 2);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const size_t li = (size_t)((y >> 2) + ((k & 2) ? dy : 0)) * latW + (x >> 2) + ((k & 1) ? dx : 0);
        if ((~mapped3[li]) & planeBit) atomicMin(&owner[li], ((uint32_t)pos << 2) | (uint32_t)k);
    }
}
template <bool EMIT>
__global__ __launch_bounds__(1024) void yk_pp_stream_kernel(PPPlanes P, PPGeo g, size_t nBits, int planeBit, const uint32_t* __restrict__ bitmap, int latW,
                                                            const uint8_t* __restrict__ mapped3, const uint32_t* __restrict__ owner, uint32_t* __restrict__ blockSums,
                                                            uint8_t* __restrict__ out) {
    __shared__ uint32_t s_tmp[32];
    const size_t pos = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const bool set = pos < nBits && ((bitmap[pos >> 5] >> (pos & 31)) & 1u);
    int x = 0, y = 0;
    uint32_t need[4] = { 0, 0, 0, 0 }, bytes = 0;
    const int dx = 1 << (g.sx - 2), dy = 1 << (g.sy - 2);
    if (set) {
        yk_pp_tile(g, (uint32_t)pos, x, y);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const size_t li = (size_t)((y >> 2) + ((k & 2) ? dy : 0)) * latW + (x >> 2) + ((k & 1) ? dx : 0);
            if (owner[li] == (((uint32_t)pos << 2) | (uint32_t)k)) { need[k] = (uint32_t)((~mapped3[li]) & planeBit); bytes += (uint32_t)__popc(need[k]); }
        }
    }
    uint32_t tot;
    const uint32_t ex = yk_block_exscan(bytes, s_tmp, &tot);
    if (!EMIT) { if (threadIdx.x == 0) blockSums[blockIdx.x] = tot; return; }
    uint32_t off = blockSums[blockIdx.x] + ex;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (!need[k]) continue;
#pragma unroll
        for (int n = 0; n < 3; n++) if ((need[k] >> n) & 1u) {
            const int v = yk_pp_r6(yk_pp_px(P, n, x + ((k & 1) ? (1 << g.sx) : 0), y + ((k & 2) ? (1 << g.sy) : 0)));
            out[off++] = (uint8_t)((v * 250 + 127) / 255);                   // CompressF(Round6(corner), colorCompressionQuad = 250) :3191, :4113-4132
        }
    }
}
__global__ __launch_bounds__(256) void yk_pp_mark_kernel(const uint32_t* __restrict__ owner, size_t lat, int planeBit, uint8_t* __restrict__ mapped3) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < lat && owner[i] != 0xFFFFFFFFu) mapped3[i] |= (uint8_t)planeBit;
}

// testOutput of FittingQuadSmooth (:3960-3971, :4096-4104): every accepted tile writes blendC6Exp, the rounded bilinear blend of the
// Round6P corners, into the preview planes of its present planes -- an encoder-side debugging aid (the decoder reconstructs from the
// Round6 corners de-quantised through CompressF, not this).  One thread per tile slot; not on any timed path.
__global__ __launch_bounds__(256) void yk_pp_preview_kernel(PPPlanes P, PPGeo g, size_t nBits, int planeBit, const uint32_t* __restrict__ bitmap, int32_t* __restrict__ preview, size_t planeElems) {
    const size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= nBits || !((bitmap[pos >> 5] >> (pos & 31)) & 1u)) return;
    int x, y; yk_pp_tile(g, (uint32_t)pos, x, y);
    const int TX = 1 << g.sx, TY = 1 << g.sy;
    if (x + TX > P.w || y + TY > P.h) return;
    for (int n = 0; n < 3; n++) {
        if (!((planeBit >> n) & 1)) continue;
        const int c0 = yk_pp_r6p(yk_pp_px(P, n, x, y)), c1 = yk_pp_r6p(yk_pp_px(P, n, x + TX, y)), c2 = yk_pp_r6p(yk_pp_px(P, n, x, y + TY)), c3 = yk_pp_r6p(yk_pp_px(P, n, x + TX, y + TY));
        for (int dy = 0; dy < TY; dy++) {
            const int tF = 1024 - dy * (1024 >> g.sy), bF = 1024 - tF;
            for (int dx = 0; dx < TX; dx++) {
                const int lF = 1024 - dx * (1024 >> g.sx), rF = 1024 - lF;
                preview[(size_t)n * planeElems + (size_t)(y + dy) * P.w + (x + dx)] = ((c0 * lF + c1 * rF) * tF + (c2 * lF + c3 * rF) * bF + ((1 << 19) - 1)) >> 20;
            }
        }
    }
}
__global__ void yk_pp_fill_i32_kernel(int32_t* __restrict__ p, size_t n, int32_t v) { const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }

// Per-plane state of the passes that follow the seven RGB passes (plane-subset gradient passes, 3-D LUT tiles): every plane starts from
// the common coverage, every lattice point an RGB pass emitted is known to all planes.  Idempotent until the next encode.
int yk_pp_activate(yk_ctx* c) {
    if (c->ppActive) return YK_OK;
    const int w = c->fullW, h = c->h, latW = w / 4 + 1, latH = h / 4 + 1;
    const size_t lat = (size_t)latW * latH, nMT = (size_t)c->mtW * c->mtH;
    // the corner lattice of the RGB passes must exist: it tells which lattice points every plane already has
    if (!c->cornersReady) { int rc = yk_launch_corners(c); if (rc) return rc; }
    c->covChStride = (nMT + 7) & ~(size_t)7;
    if (!c->covCh) {
        YK_HIP(c, hipMalloc(&c->covCh, 3 * c->covChStride * sizeof(uint16_t) + 16));
        YK_HIP(c, hipMalloc(&c->mapped3, lat + 16));
    }
    YK_HIP(c, hipMemsetAsync(c->covCh, 0, 3 * c->covChStride * sizeof(uint16_t) + 16, c->stream));
    const size_t n = nMT > lat ? nMT : lat;
    hipLaunchKernelGGL(yk_pp_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->coverage, nMT, c->covCh, c->covChStride,
                       c->latticeOwner, lat, c->mapped3);
    YK_HIP(c, hipGetLastError());
    c->ppActive = true;
    return YK_OK;
}

extern "C" {

int yk_gradient_partial_pass(yk_ctx* c, int rejectFactor, int planeBit, int sx, int sy, int* tilesAccepted) {
    if (!c) return YK_ERR_BAD_ARG;
    if (!c->encoded) return yk_fail(c, YK_ERR_STATE, "yk_encode_tiles first (the partial-plane passes follow the seven RGB passes)");
    if (planeBit < 1 || planeBit > 7) return yk_fail(c, YK_ERR_BAD_ARG, "planeBit must be 1..7");
    if (rejectFactor < 0 || rejectFactor > 64) return yk_fail(c, YK_ERR_BAD_ARG, "rejectFactor out of range");
    static const int ok[7][2] = { {4,4},{4,3},{3,4},{3,3},{3,2},{2,3},{2,2} };
    bool found = false; for (auto& o : ok) found |= (o[0] == sx && o[1] == sy);
    if (!found) return yk_fail(c, YK_ERR_BAD_ARG, "unsupported tile format");
    if (c->nFrames != 1 || c->y0 != 0 || c->h != c->fullH) return yk_fail(c, YK_ERR_STATE, "partial-plane passes work on single whole images (no batch, no stripe)");
    YK_HIP(c, hipSetDevice(c->device));
    const int w = c->fullW, h = c->h, latW = w / 4 + 1, latH = h / 4 + 1;
    const size_t lat = (size_t)latW * latH, nMT = (size_t)c->mtW * c->mtH;
    { int rc = yk_pp_activate(c); if (rc) return rc; }
    const PPGeo g = yk_pp_geo(sx, sy, w);
    const size_t nBits = (size_t)g.xBB * ((h + g.bigY - 1) / g.bigY) * g.bitCount, nWords = (nBits + 31) / 32, nb = (nBits + 1023) / 1024;
    const size_t needScratch = lat + nb + 64;
    if (c->ppBitmapCap < nWords * 4) { if (c->ppBitmap) (void)hipFree(c->ppBitmap); YK_HIP(c, hipMalloc(&c->ppBitmap, nWords * 4 + 16)); c->ppBitmapCap = nWords * 4; }
    if (c->ppScratchElems < needScratch) { if (c->ppScratch) (void)hipFree(c->ppScratch); YK_HIP(c, hipMalloc(&c->ppScratch, needScratch * 4)); c->ppScratchElems = needScratch; }
    if (c->ppStreamCap < lat * 3 + 16) { if (c->ppStream) (void)hipFree(c->ppStream); YK_HIP(c, hipMalloc(&c->ppStream, lat * 3 + 16)); c->ppStreamCap = lat * 3 + 16; }
    uint32_t* owner = c->ppScratch; uint32_t* blockSums = owner + lat; uint32_t* total = blockSums + nb; uint32_t* accepted = total + 1;
    YK_HIP(c, hipMemsetAsync(c->ppBitmap, 0, nWords * 4, c->stream));
    YK_HIP(c, hipMemsetAsync(owner, 0xFF, lat * 4, c->stream));
    YK_HIP(c, hipMemsetAsync(total, 0, 8, c->stream));
    PPPlanes P; for (int n = 0; n < 3; n++) P.p[n] = c->plane[n];
    P.strideElems = c->strideElems; P.w = w; P.h = h; P.hAvail = h + c->halo;
    const unsigned g256 = (unsigned)((nBits + 255) / 256);
    hipLaunchKernelGGL(yk_pp_decide_kernel, dim3(g256), dim3(256), 0, c->stream, P, g, nBits, planeBit, rejectFactor, c->covCh, c->covChStride, c->mtW, c->ppBitmap, accepted);
    hipLaunchKernelGGL(yk_pp_paint_kernel, dim3(g256), dim3(256), 0, c->stream, g, nBits, planeBit, c->ppBitmap, reinterpret_cast<uint32_t*>(c->covCh), c->covChStride,
                       reinterpret_cast<uint32_t*>(c->coverage), c->mtW);
    hipLaunchKernelGGL(yk_pp_owner_kernel, dim3(g256), dim3(256), 0, c->stream, g, nBits, planeBit, c->ppBitmap, latW, c->mapped3, owner);
    hipLaunchKernelGGL(yk_pp_stream_kernel<false>, dim3((unsigned)nb), dim3(1024), 0, c->stream, P, g, nBits, planeBit, c->ppBitmap, latW, c->mapped3, owner, blockSums, (uint8_t*)nullptr);
    hipLaunchKernelGGL(yk_u32_scanblocks_kernel, dim3(1), dim3(1024), 0, c->stream, blockSums, (int)nb, total);
    hipLaunchKernelGGL(yk_pp_stream_kernel<true>, dim3((unsigned)nb), dim3(1024), 0, c->stream, P, g, nBits, planeBit, c->ppBitmap, latW, c->mapped3, owner, blockSums, c->ppStream);
    hipLaunchKernelGGL(yk_pp_mark_kernel, dim3((unsigned)((lat + 255) / 256)), dim3(256), 0, c->stream, owner, lat, planeBit, c->mapped3);
    YK_HIP(c, hipGetLastError());
    uint32_t res[2];
    YK_HIP(c, hipMemcpyAsync(res, total, sizeof res, hipMemcpyDeviceToHost, c->stream));
    YK_HIP(c, hipStreamSynchronize(c->stream));
    c->ppStreamBytes = res[0]; c->ppAccepted = (int)res[1];
    c->ppLastBit = planeBit; c->ppLastSx = sx; c->ppLastSy = sy;
    c->r1Ready = false;                                                      // the 1-D path now has less to code
    if (tilesAccepted) *tilesAccepted = c->ppAccepted;
    // the byte size of this pass's bitmap as the reference allocates it (:3770-3777)
    c->ppBitmapBytes = (size_t)(nBits >> 3);
    return YK_OK;
}

int yk_gradient_preview(yk_ctx* c, int pass, int32_t* hostOut, size_t capElems) {
    if (!c || pass < 0 || pass > 7) return YK_ERR_BAD_ARG;
    if (!c->encoded) return yk_fail(c, YK_ERR_STATE, "yk_encode_tiles first");
    if (c->nFrames != 1 || c->y0 != 0 || c->h != c->fullH) return yk_fail(c, YK_ERR_STATE, "the preview planes exist for single whole images");
    if (pass == 7 && (!c->ppActive || !c->ppBitmap || c->ppLastBit == 0)) return yk_fail(c, YK_ERR_STATE, "no plane-subset pass ran since the encode");
    YK_HIP(c, hipSetDevice(c->device));
    const int w = c->fullW, h = c->h;
    const size_t planeElems = (size_t)w * h;
    if (!c->preview) { YK_HIP(c, hipMalloc(&c->preview, 3 * planeElems * sizeof(int32_t))); c->previewFresh = false; }
    if (!c->previewFresh) {                                                  // a new encode starts from untouched planes (INT32_MIN = no tile wrote here)
        hipLaunchKernelGGL(yk_pp_fill_i32_kernel, dim3((unsigned)((3 * planeElems + 255) / 256)), dim3(256), 0, c->stream, c->preview, 3 * planeElems, (int32_t)0x80000000);
        c->previewFresh = true;
    }
    static const int shp[7][2] = { {4,4},{4,3},{3,4},{3,3},{3,2},{2,3},{2,2} };
    const int sx = pass < 7 ? shp[pass][0] : c->ppLastSx, sy = pass < 7 ? shp[pass][1] : c->ppLastSy, planeBit = pass < 7 ? 7 : c->ppLastBit;
    const PPGeo g = yk_pp_geo(sx, sy, w);
    const size_t nBits = (size_t)g.xBB * ((h + g.bigY - 1) / g.bigY) * g.bitCount;
    PPPlanes P; for (int n = 0; n < 3; n++) P.p[n] = c->plane[n];
    P.strideElems = c->strideElems; P.w = w; P.h = h; P.hAvail = h + c->halo;
    const uint32_t* bm = pass < 7 ? reinterpret_cast<const uint32_t*>(c->bitmap[pass]) : c->ppBitmap;
    if (pass < 7 && (c->bitmapBytes[pass] & 3)) YK_HIP(c, hipMemsetAsync(c->bitmap[pass] + c->bitmapBytes[pass], 0, 4 - (c->bitmapBytes[pass] & 3), c->stream));
    hipLaunchKernelGGL(yk_pp_preview_kernel, dim3((unsigned)((nBits + 255) / 256)), dim3(256), 0, c->stream, P, g, nBits, planeBit, bm, c->preview, planeElems);
    YK_HIP(c, hipGetLastError());
    if (hostOut) {
        if (capElems < 3 * planeElems) return yk_fail(c, YK_ERR_RANGE, "preview buffer too small (3 planes of w*h int32)");
        YK_HIP(c, hipMemcpyAsync(hostOut, c->preview, 3 * planeElems * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    }
    YK_HIP(c, hipStreamSynchronize(c->stream));
    return YK_OK;
}

int yk_partial_bitmap(yk_ctx* c, uint8_t* hostOut, size_t cap, size_t* nBytes) {
    if (!c) return YK_ERR_BAD_ARG;
    if (!c->ppBitmap) return yk_fail(c, YK_ERR_STATE, "yk_gradient_partial_pass first");
    if (nBytes) *nBytes = c->ppBitmapBytes;
    if (hostOut) {
        if (cap < c->ppBitmapBytes) return yk_fail(c, YK_ERR_RANGE, "bitmap buffer too small");
        YK_HIP(c, hipSetDevice(c->device));
        YK_HIP(c, hipMemcpyAsync(hostOut, c->ppBitmap, c->ppBitmapBytes, hipMemcpyDeviceToHost, c->stream));
        YK_HIP(c, hipStreamSynchronize(c->stream));
    }
    return YK_OK;
}

int yk_partial_corners(yk_ctx* c, uint8_t* hostOut, size_t cap, size_t* nBytes) {
    if (!c) return YK_ERR_BAD_ARG;
    if (!c->ppStream) return yk_fail(c, YK_ERR_STATE, "yk_gradient_partial_pass first");
    if (nBytes) *nBytes = c->ppStreamBytes;
    if (hostOut && c->ppStreamBytes) {
        if (cap < c->ppStreamBytes) return yk_fail(c, YK_ERR_RANGE, "corner buffer too small");
        YK_HIP(c, hipSetDevice(c->device));
        YK_HIP(c, hipMemcpyAsync(hostOut, c->ppStream, c->ppStreamBytes, hipMemcpyDeviceToHost, c->stream));
        YK_HIP(c, hipStreamSynchronize(c->stream));
    }
    return YK_OK;
}

int yk_coverage_plane(yk_ctx* c, int plane, uint16_t* hostOut, size_t capElems) {
    if (!c || !hostOut || plane < 0 || plane > 2) return YK_ERR_BAD_ARG;
    if (!c->encoded) return yk_fail(c, YK_ERR_STATE, "yk_encode_tiles first");
    const size_t n = (size_t)c->mtW * c->mtH;
    if (capElems < n) return yk_fail(c, YK_ERR_RANGE, "coverage buffer too small");
    YK_HIP(c, hipSetDevice(c->device));
    const uint16_t* src = c->ppActive ? c->covCh + (size_t)plane * c->covChStride : c->coverage;      // before any partial pass the planes agree
    YK_HIP(c, hipMemcpyAsync(hostOut, src, n * sizeof(uint16_t), hipMemcpyDeviceToHost, c->stream));
    YK_HIP(c, hipStreamSynchronize(c->stream));
    return YK_OK;
}

}  // extern "C"
