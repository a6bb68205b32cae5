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

// (n + 0.5) * rcp(d) truncated == n / d for 0 <= n < 4096, 1 <= d <= 255 (exhaustive: yk_selftest 2)
__device__ __forceinline__ int yk_r1_div(int n, int d) { return __float2int_rz(((float)n + 0.5f) * __builtin_amdgcn_rcpf((float)d)); }

// One wave64 per workgroup = a 64x16 strip = 16 tiles; a lane owns a 4x4 cell = one quadrant of an 8x8 tile, a tile is the four
// lanes {l, l^1, l^4, l^5} (same geometry as yk_encode2_kernel).  Per plane: 256-byte row segments are loaded with 16-byte
// loads and parked in LDS as bytes; the 256-bin histogram of a tile lives in 64 LDS words (4 byte-wide counters per word,
// counts <= 64), filled with ds_add and read back per pixel; min / max / mode reductions are two xor-shuffles.
// DIRECT: the stream offsets of every tile are known before the kernel runs (they depend on the coverage alone: yk_r1_offsets_kernel + one scan),
// so pixel bytes and parameters go straight to their place in the reference's streams -- no per-tile slots, no compaction pass behind it.
// CACHED: the pixels come from the fused kernel's pixel cache (yk_set_pixel_cache: packed 0x00BBGGRR words of the uncovered cells, 4 B per pixel,
// [strip][cell row][lane in that kernel's Morton order]) instead of the int32 planes (12 B per pixel): a lane's sixteen pixels are four 16-byte
// loads, held in registers across the three planes -- no staging through LDS, and every input sample of the whole path is read from HBM once.
struct R1Direct { const uint32_t* offInBlk; const uint32_t* blockT; const uint32_t* blockP; const uint32_t* totals; uint8_t* pixOut; uint8_t* typeOut; const uint4* pixCache; };
template <bool DIRECT, bool CACHED = false>
__global__ __launch_bounds__(64) void yk_range1d_kernel(const int32_t* __restrict__ pR, const int32_t* __restrict__ pG, const int32_t* __restrict__ pB,
                                                        int strideElems, int w, int h, const uint16_t* __restrict__ coverage, int mtW,
                                                        int tilesW, size_t T8, uint8_t* __restrict__ slots, uint8_t* __restrict__ params,
                                                        uint32_t* __restrict__ cntTiles, uint32_t* __restrict__ cntPix, int onlyPlane, const R1Direct D) {
    // onlyPlane < 0: the three planes share `coverage` (no partial-plane pass ran).  Otherwise this launch codes plane `onlyPlane` alone
    // against that plane's own coverage (mapSmoothTile->GetPlane(p), EncoderContext.cpp:9451-9465).
    __shared__ __attribute__((aligned(16))) uint32_t s_px[16 * 16];          // 16 rows x 64 pixels, one byte each
    // per tile: 256 byte-wide bins in 64 words; the tiles' histograms are 65 words apart: neighbouring tiles hold similar values, and with a
    // stride of 64 the same bin of every tile sat in one LDS bank (the sixteen ds_add of a lane then serialised across the whole wave)
    __shared__ __attribute__((aligned(16))) uint32_t s_hist[16 * 65 + 16];
    const int lane = threadIdx.x;
    const int xBB = (w + 63) >> 6;
    const int BX = (int)(blockIdx.x % xBB), SY = (int)(blockIdx.x / xBB);    // strip coordinates (64 px, 16 rows)
    const int q = lane >> 4, cell = lane & 15, cx = cell & 3, cy = cell >> 2;
    const int gxCell = BX * 64 + q * 16 + cx * 4, gyCell = SY * 16 + cy * 4;
    const bool mtIn = (BX * 64 + q * 16 < w) && (SY * 16 < h);
    const uint32_t cw = mtIn ? coverage[(size_t)SY * mtW + (BX * 4 + q)] : 0xFFFFu;
    const int cxl = cx & 1, cyl = cy & 1;
    const int tgx = gxCell - cxl * 4, tgy = gyCell - cyl * 4;
    const bool tileIn = (tgx + 8 <= w) && (tgy + 8 <= h);                    // the reference iterates whole 8x8 tiles (w, h multiples of 8)
    const int c00 = (cy & 2) * 4 + (cx & 2);                                 // bit of the tile's top-left cell in the coverage word
    const bool u00 = tileIn && !((cw >> c00) & 1), u10 = tileIn && !((cw >> (c00 + 1)) & 1);
    const bool u01 = tileIn && !((cw >> (c00 + 4)) & 1), u11 = tileIn && !((cw >> (c00 + 5)) & 1);
    const bool valid = cyl ? (cxl ? u11 : u01) : (cxl ? u10 : u00);          // this lane's quadrant is uncovered (:8420-8431)
    const int nTop = (int)u00 + (int)u10, nBot = (int)u01 + (int)u11;
    const int nPix = 16 * (nTop + nBot);
    const size_t ti = (size_t)(tgy >> 3) * tilesW + (tgx >> 3);
    const bool writer = tileIn && cxl == 0 && cyl == 0;
    if (!DIRECT && writer) { cntTiles[ti] = nPix ? 1u : 0u; cntPix[ti] = (uint32_t)nPix; }
    const unsigned long long validCells = __ballot(valid);                   // bit = lane = macroTile * 16 + cellY * 4 + cellX
    if (validCells == 0ULL) return;
    // ---- at most four cells to code (a smooth strip's last column of cells next to other content): the sixteen lanes of a quarter of the wave take one
    // pixel each of one of those cells instead of 60 lanes watching four walk sixteen pixels; the per-tile reductions go through LDS atomics.
    const int nValid = __popcll(validCells);
    if (nValid <= 4) {
        __shared__ uint8_t s_vl[4];
        __shared__ int s_red[16][3];                                         // per tile: mode key, min, max of what is left
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(validCells >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)validCells, 0u));
        if (valid) s_vl[rank] = (uint8_t)lane;
        __syncthreads();
        const int si = lane >> 4, k = lane & 15;
        const bool act = si < nValid;
        const int src = (int)s_vl[act ? si : 0];
        const int sq = src >> 4, scx = src & 3, scy = (src >> 2) & 3, scxl = scx & 1, scyl = scy & 1;
        const uint32_t cws = coverage[(size_t)SY * mtW + (BX * 4 + sq)];
        const int sc00 = (scy & 2) * 4 + (scx & 2);
        const int s00 = !((cws >> sc00) & 1), s10 = !((cws >> (sc00 + 1)) & 1), s01 = !((cws >> (sc00 + 4)) & 1), s11 = !((cws >> (sc00 + 5)) & 1);
        const int snTop = s00 + s10, snBot = s01 + s11;
        const int sposBase = scyl ? (16 * snTop + (scxl ? 4 * s01 : 0)) : (scxl ? 4 * s00 : 0), sposStep = 4 * (scyl ? snBot : snTop);
        const int sgx = BX * 64 + sq * 16 + scx * 4, sgy = SY * 16 + scy * 4;          // the cell's origin
        const size_t sti = (size_t)((sgy - scyl * 4) >> 3) * tilesW + ((sgx - scxl * 4) >> 3);
        const int stw = sq * 4 + (scy >> 1) * 2 + (scx >> 1);
        uint32_t* const shist = &s_hist[stw * 65];
        const int32_t* planesS[3] = { pR, pG, pB };
        uint32_t spw = 0;                                                    // CACHED: pixel k of the source cell, the three planes in one word
        if (CACHED && act) {
            const int sm = sq * 16 + (scy >> 1) * 8 + (scx >> 1) * 4 + (scy & 1) * 2 + (scx & 1);      // the source cell's lane in the fused kernel's order
            spw = reinterpret_cast<const uint32_t*>(D.pixCache + ((size_t)blockIdx.x * 4 + (k >> 2)) * 64 + sm)[k & 3];
        }
        for (int p = (onlyPlane < 0 ? 0 : onlyPlane); p < (onlyPlane < 0 ? 3 : onlyPlane + 1); p++) {
            int v = 0;                                                           // CompressF(v,255) == v
            if (CACHED) v = (int)((spw >> (8 * p)) & 255u);
            else if (act) v = planesS[p][(size_t)(sgy + (k >> 2)) * strideElems + sgx + (k & 3)] & 255;
            __syncthreads();
            // lane-contiguous 16-byte stores (a lane clearing its own 64-byte run put the whole wave on two LDS banks)
            *reinterpret_cast<uint4*>(&s_hist[lane * 4]) = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(&s_hist[256 + lane * 4]) = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(&s_hist[512 + lane * 4]) = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(&s_hist[768 + lane * 4]) = make_uint4(0, 0, 0, 0);
            if (lane < 4) *reinterpret_cast<uint4*>(&s_hist[1024 + lane * 4]) = make_uint4(0, 0, 0, 0);
            if (lane < 16) { s_red[lane][0] = -1; s_red[lane][1] = 99999; s_red[lane][2] = -99999; }
            __syncthreads();
            if (act) atomicAdd(&shist[v >> 2], 1u << (8 * (v & 3)));
            __syncthreads();
            if (act) atomicMax(&s_red[stw][0], (int)((((shist[v >> 2] >> (8 * (v & 3))) & 255u) << 8) | (uint32_t)v));   // right-most mode (:8335-8356)
            __syncthreads();
            int color0 = s_red[stw][0] & 255;
            if (color0 == 0) color0 = 1;
            if (color0 == 255) color0 = 254;
            const bool isC0 = (v >= color0 - 1) && (v <= color0 + 1);
            if (act && !isC0) { atomicMin(&s_red[stw][1], v); atomicMax(&s_red[stw][2], v); }                              // Model1 (:8358-8381)
            __syncthreads();
            const int mn = s_red[stw][1], mx = s_red[stw][2];
            int minCol = 0, delta = 0;
            if (mn != 99999) { minCol = mn; delta = mx - mn; }
            if (act) {
                int out = 0;
                if (!isC0) {
                    int idx = 0;
                    if (delta) { const int n = (v - minCol) * 15 + (delta >> 1) - 1; idx = n < 0 ? -1 : yk_r1_div(n, delta); }      // GetValueModel1 (:8383-8391)
                    out = 1 + idx;
                }
                if (DIRECT) {
                    const uint32_t e = D.offInBlk[sti], blk = (uint32_t)(sti >> 10);
                    D.pixOut[(size_t)p * D.totals[1] + D.blockP[blk] + (e >> 11) + sposBase + (k >> 2) * sposStep + (k & 3)] = (uint8_t)out;
                    if (k == 0) {                                            // every coded cell of a tile writes the same three parameters
                        uint8_t* qp = D.typeOut + ((size_t)p * D.totals[0] + D.blockT[blk] + (e & 2047u)) * 3;
                        qp[0] = (uint8_t)color0; qp[1] = (uint8_t)minCol; qp[2] = (uint8_t)delta;
                    }
                } else {
                    slots[((size_t)p * T8 + sti) * 64 + sposBase + (k >> 2) * sposStep + (k & 3)] = (uint8_t)out;
                    if (k == 0) {                                            // every coded cell of a tile writes the same three parameters
                        uint8_t* qp = params + ((size_t)p * T8 + sti) * 4;
                        qp[0] = (uint8_t)color0; qp[1] = (uint8_t)minCol; qp[2] = (uint8_t)delta;
                    }
                }
            }
        }
        return;
    }
    const int tw = q * 4 + (cy >> 1) * 2 + (cx >> 1);                        // tile index inside the strip
    uint32_t* hist = &s_hist[tw * 65];
    // position of the lane's pixel rows among the tile's emitted pixels: top half rows (left then right quadrant), then bottom (:8420-8453)
    const int posBase = cyl ? (16 * nTop + (cxl ? 4 * (int)u01 : 0)) : (cxl ? 4 * (int)u00 : 0);
    const int posStep = 4 * (cyl ? nBot : nTop);
    const int32_t* planes[3] = { pR, pG, pB };
    const int g4 = (lane & 15) * 4, r0 = lane >> 4;
    uint4 cpx[4] = {};
    if (CACHED && valid) {
        const int m = q * 16 + (cy >> 1) * 8 + (cx >> 1) * 4 + (cy & 1) * 2 + (cx & 1);              // this cell's lane in the fused kernel's order
        const uint4* src = D.pixCache + (size_t)blockIdx.x * 256 + m;
#pragma unroll
        for (int r = 0; r < 4; r++) cpx[r] = src[r * 64];
    }
    for (int p = (onlyPlane < 0 ? 0 : onlyPlane); p < (onlyPlane < 0 ? 3 : onlyPlane + 1); p++) {
        uint32_t row[4];
        if (CACHED) {
            // byte p of the row's four pixel words
            const uint32_t sel = 0x0C0C0000u | ((4u + (uint32_t)p) << 8) | (uint32_t)p;
#pragma unroll
            for (int r = 0; r < 4; r++)
                row[r] = __builtin_amdgcn_perm(__builtin_amdgcn_perm(cpx[r].w, cpx[r].z, sel), __builtin_amdgcn_perm(cpx[r].y, cpx[r].x, sel), 0x05040100u);
            __syncthreads();                                                 // previous plane's readers of the histograms are done (single wave: LDS fence)
            *reinterpret_cast<uint4*>(&s_hist[lane * 4]) = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(&s_hist[256 + lane * 4]) = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(&s_hist[512 + lane * 4]) = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(&s_hist[768 + lane * 4]) = make_uint4(0, 0, 0, 0);
            if (lane < 4) *reinterpret_cast<uint4*>(&s_hist[1024 + lane * 4]) = make_uint4(0, 0, 0, 0);
            __syncthreads();
        } else
        // ---- stage the strip of this plane as bytes -------------------------------------------------------------
        {
            const int gx = BX * 64 + g4;
            uint32_t pk[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int gy = SY * 16 + r0 + 4 * k;
                int4 v = make_int4(0, 0, 0, 0);
                // only the row segments of cells that are coded are fetched (a smooth strip next to a contour has 4 of its 64 cells left:
                // 768 bytes instead of 12 KB); segment (lane & 15) of row r0 + 4k lies in cell (x = lane & 3, y = k) of macro-tile (lane >> 2) & 3
                const bool need = (validCells >> ((((lane >> 2) & 3) << 4) + (k << 2) + (lane & 3))) & 1ULL;
                if (need && gx + 3 < w && gy < h) v = *reinterpret_cast<const int4*>(planes[p] + (size_t)gy * strideElems + gx);
                pk[k] = ((uint32_t)v.x & 255u) | (((uint32_t)v.y & 255u) << 8) | (((uint32_t)v.z & 255u) << 16) | ((uint32_t)v.w << 24);   // CompressF(v,255) == v
            }
            __syncthreads();                                                 // previous plane's readers are done (single wave: LDS fence)
#pragma unroll
            for (int k = 0; k < 4; k++) s_px[(r0 + 4 * k) * 16 + (lane & 15)] = pk[k];
            // lane-contiguous 16-byte stores (a lane clearing its own 64-byte run put the whole wave on two LDS banks)
            *reinterpret_cast<uint4*>(&s_hist[lane * 4]) = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(&s_hist[256 + lane * 4]) = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(&s_hist[512 + lane * 4]) = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(&s_hist[768 + lane * 4]) = make_uint4(0, 0, 0, 0);
            if (lane < 4) *reinterpret_cast<uint4*>(&s_hist[1024 + lane * 4]) = make_uint4(0, 0, 0, 0);
            __syncthreads();
        }
        if (!CACHED) {
#pragma unroll
            for (int r = 0; r < 4; r++) row[r] = s_px[(cy * 4 + r) * 16 + q * 4 + cx];
        }
        // ---- histogram of the tile's valid pixels, right-most mode (FindAndRemoveMostUsedColor, :8335-8356) --------
        // bin v is byte v of the tile's 64 words: the add goes to its word, the count comes back with one byte read
        uint8_t* const histB = reinterpret_cast<uint8_t*>(hist);
        if (valid) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t v = (row[k >> 2] >> (8 * (k & 3))) & 255u;
                atomicAdd(reinterpret_cast<uint32_t*>(histB + (v & 0xFCu)), 1u << ((v << 3) & 31u));
            }
        }
        __syncthreads();
        int key = -1;
        if (valid) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t v = (row[k >> 2] >> (8 * (k & 3))) & 255u;
                key = max(key, (int)(((uint32_t)histB[v] << 8) | v));
            }
        }
        key = max(key, __shfl_xor(key, 1)); key = max(key, __shfl_xor(key, 4));
        int color0 = key & 255;
        if (color0 == 0) color0 = 1;
        if (color0 == 255) color0 = 254;
        // ---- Model1 over what is left (:8358-8381) ------------------------------------------------------------------
        const uint32_t c0lo = (uint32_t)(color0 - 1);                         // color0 +- 1  <=>  v - c0lo <= 2 (unsigned)
        int mn = 99999, mx = -99999;
        if (valid) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int v = (int)((row[k >> 2] >> (8 * (k & 3))) & 255u);
                const bool isC0 = (uint32_t)v - c0lo <= 2u;
                mn = isC0 ? mn : min(mn, v); mx = isC0 ? mx : max(mx, v);
            }
        }
        mn = min(mn, __shfl_xor(mn, 1)); mn = min(mn, __shfl_xor(mn, 4));
        mx = max(mx, __shfl_xor(mx, 1)); mx = max(mx, __shfl_xor(mx, 4));
        int minCol = 0, delta = 0;
        if (mn != 99999) { minCol = mn; delta = mx - mn; }
        if (valid) {
            uint8_t* slot;
            if (DIRECT) { const uint32_t e = D.offInBlk[ti]; slot = D.pixOut + (size_t)p * D.totals[1] + D.blockP[ti >> 10] + (e >> 11); }
            else slot = slots + ((size_t)p * T8 + ti) * 64;
            // GetValueModel1 (:8383-8391) as one multiply-add and a shift per pixel (yk_r1_magic: exact, yk_selftest 4); no branch per pixel
            uint32_t A, B;
            yk_r1_magic(delta, minCol, &A, &B);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                uint32_t o4 = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t v = (row[r] >> (8 * i)) & 255u;
                    const uint32_t out = (v - c0lo <= 2u) ? 0u : (__umul24(v, A) + B) >> 20;
                    o4 |= out << (8 * i);
                }
                *reinterpret_cast<uint32_t*>(slot + posBase + r * posStep) = o4;
            }
        }
        if (writer && nPix) {
            uint8_t* qp;
            if (DIRECT) qp = D.typeOut + ((size_t)p * D.totals[0] + D.blockT[ti >> 10] + (D.offInBlk[ti] & 2047u)) * 3;
            else qp = params + ((size_t)p * T8 + ti) * 4;
            qp[0] = (uint8_t)color0; qp[1] = (uint8_t)minCol; qp[2] = (uint8_t)delta;
        }
    }
}

// Stream offsets from the coverage alone (thread = tile, 1024 tiles per workgroup): a tile emits 16 bytes per uncovered quadrant and one
// parameter triple when it emits anything.  Packed like the decoder's scan: coded tiles in the low 11 bits, pixel bytes above.
__global__ __launch_bounds__(1024) void yk_r1_offsets_kernel(const uint16_t* __restrict__ coverage, int mtW, int tilesW, size_t T8,
                                                             uint32_t* __restrict__ offInBlk, uint32_t* __restrict__ blockT, uint32_t* __restrict__ blockP) {
    __shared__ uint32_t s_tmp[32];
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    uint32_t cnt = 0;
    if (i < T8) {
        const int tx = (int)(i % tilesW), ty = (int)(i / tilesW);
        const uint32_t cw = coverage[(size_t)(ty >> 1) * mtW + (tx >> 1)];
        const int c00 = (ty & 1) * 8 + (tx & 1) * 2;
        const uint32_t n = 4u - (((cw >> c00) & 1u) + ((cw >> (c00 + 1)) & 1u) + ((cw >> (c00 + 4)) & 1u) + ((cw >> (c00 + 5)) & 1u));
        cnt = (n ? 1u : 0u) | ((16u * n) << 11);
    }
    uint32_t tot;
    const uint32_t e = yk_block_exscan(cnt, s_tmp, &tot);
    if (i < T8) offInBlk[i] = e;
    if (threadIdx.x == 0) { blockT[blockIdx.x] = tot & 2047u; blockP[blockIdx.x] = tot >> 11; }
}
// both block-sum arrays -> exclusive prefixes in place, totals[0] = coded tiles, totals[1] = pixel bytes of ONE plane
__global__ __launch_bounds__(1024) void yk_r1_scan_kernel(uint32_t* __restrict__ blockT, uint32_t* __restrict__ blockP, int nBlocks, uint32_t* __restrict__ totals) {
    __shared__ uint32_t s_tmp[32];
#pragma unroll 1
    for (int a = 0; a < 2; a++) {
        uint32_t* arr = a ? blockP : blockT;
        uint32_t base = 0;
        for (int start = 0; start < nBlocks; start += 1024) {
            const int i = start + threadIdx.x;
            const uint32_t v = i < nBlocks ? arr[i] : 0u;
            uint32_t tot;
            const uint32_t e = yk_block_exscan(v, s_tmp, &tot);
            if (i < nBlocks) arr[i] = base + e;
            base += tot;
        }
        if (threadIdx.x == 0) totals[a] = base;
    }
}

// One workgroup = 1024 consecutive tiles of one plane: scans give every tile its stream offsets, then 16 lanes per tile copy
// its pixel bytes as 4-byte words (offsets are multiples of 16) and one lane its three parameter bytes.
__global__ __launch_bounds__(1024) void yk_range1d_pack_kernel(const uint32_t* __restrict__ cntTiles, const uint32_t* __restrict__ cntPix,
                                                               const uint32_t* __restrict__ baseTiles, const uint32_t* __restrict__ basePix,
                                                               const uint32_t* __restrict__ totals, size_t T8, const uint8_t* __restrict__ slots,
                                                               const uint8_t* __restrict__ params, uint8_t* __restrict__ pixOut, uint8_t* __restrict__ typeOut,
                                                               int planeOverride, const uint32_t* __restrict__ runBase) {
    __shared__ uint32_t s_tmp[32];
    __shared__ uint32_t s_offT[1024], s_offP[1024], s_n[1024];
    const int p = planeOverride < 0 ? (int)blockIdx.y : planeOverride;
    const size_t baseT = planeOverride < 0 ? (size_t)p * totals[0] : (size_t)runBase[0], baseP = planeOverride < 0 ? (size_t)p * totals[1] : (size_t)runBase[1];
    const size_t i0 = (size_t)blockIdx.x * 1024;
    {
        const size_t i = i0 + threadIdx.x;
        uint32_t tot;
        const uint32_t ct = i < T8 ? cntTiles[i] : 0u, cp = i < T8 ? cntPix[i] : 0u;
        const uint32_t et = yk_block_exscan(ct, s_tmp, &tot);
        const uint32_t ep = yk_block_exscan(cp, s_tmp, &tot);
        s_offT[threadIdx.x] = baseTiles[blockIdx.x] + et; s_offP[threadIdx.x] = basePix[blockIdx.x] + ep; s_n[threadIdx.x] = cp;
    }
    __syncthreads();
    // every tile holds a multiple of 16 bytes and all offsets are multiples of 16: four lanes per tile copy 16-byte pieces
    const int piece = threadIdx.x & 3;
    for (int it = 0; it < 4; it++) {
        const int t = it * 256 + (threadIdx.x >> 2);
        const size_t i = i0 + t;
        if (i >= T8) break;
        const uint32_t n = s_n[t];
        if (!n) continue;
        const size_t po = baseP + s_offP[t];
        if ((uint32_t)piece * 16 < n)
            *reinterpret_cast<uint4*>(pixOut + po + piece * 16) = *reinterpret_cast<const uint4*>(slots + ((size_t)p * T8 + i) * 64 + piece * 16);
        if (piece == 0) {
            const size_t to = (baseT + s_offT[t]) * 3;
            const uint8_t* qp = params + ((size_t)p * T8 + i) * 4;
            typeOut[to] = qp[0]; typeOut[to + 1] = qp[1]; typeOut[to + 2] = qp[2];
        }
    }
}

__global__ void yk_r1_next_plane_kernel(uint32_t* __restrict__ runBase, const uint32_t* __restrict__ tot) { if (threadIdx.x < 2) runBase[2 + threadIdx.x] = runBase[threadIdx.x] + tot[threadIdx.x]; }

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
    const unsigned nStrips = (unsigned)(((c->fullW + 63) / 64) * ((c->h + 15) / 16));
    uint32_t t[2];
    if (!c->ppActive) {
        // three launches: offsets of every tile from the coverage, one scan, the coder writing straight into the streams (before: coder into
        // per-tile slots, two block sums, two scans, a compaction pass re-reading the slots)
        { int rc = yk_stage_begin(c, YK_STAGE_RANGE1D_PACK); if (rc) return rc; }
        hipLaunchKernelGGL(yk_r1_offsets_kernel, dim3((unsigned)nb), dim3(1024), 0, c->stream, c->coverage, c->mtW, c->tilesW, T8, cT, bT, bP);
        hipLaunchKernelGGL(yk_r1_scan_kernel, dim3(1), dim3(1024), 0, c->stream, bT, bP, (int)nb, tot);
        { int rc = yk_stage_end(c, YK_STAGE_RANGE1D_PACK); if (rc) return rc; }
        { int rc = yk_stage_begin(c, YK_STAGE_RANGE1D); if (rc) return rc; }
        R1Direct D; D.offInBlk = cT; D.blockT = bT; D.blockP = bP; D.totals = tot; D.pixOut = c->r1Pix; D.typeOut = c->r1Type; D.pixCache = c->pixCache;
        if (c->pixCacheValid && c->pixCache)                                  // the fused kernel left the uncovered cells' pixels: 4 B per pixel instead of 12
            hipLaunchKernelGGL((yk_range1d_kernel<true, true>), dim3(nStrips), dim3(64), 0, c->stream, c->plane[0], c->plane[1], c->plane[2], c->strideElems,
                               c->fullW, c->h, c->coverage, c->mtW, c->tilesW, T8, (uint8_t*)nullptr, (uint8_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, -1, D);
        else
            hipLaunchKernelGGL(yk_range1d_kernel<true>, dim3(nStrips), dim3(64), 0, c->stream, c->plane[0], c->plane[1], c->plane[2], c->strideElems,
                               c->fullW, c->h, c->coverage, c->mtW, c->tilesW, T8, (uint8_t*)nullptr, (uint8_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, -1, D);
        YK_HIP(c, hipGetLastError());
        { int rc = yk_stage_end(c, YK_STAGE_RANGE1D); if (rc) return rc; }
        // the totals stay on the device until a getter needs them (yk_range1d_finish): callers that keep frames in flight are not stopped here
        c->r1TotalsDev = tot; c->r1TotalsPending = true; c->r1Ready = true;
        return YK_OK;
    } else {
        // partial-plane passes ran: every plane has its own coverage, so its own counts, scans and stream offsets
        uint32_t* runBase = tot + 4;                                         // [4][2]: (tile-planes, pixel bytes) before plane p; [3] = totals
        YK_HIP(c, hipMemsetAsync(runBase, 0, 2 * sizeof(uint32_t), c->stream));
        for (int p = 0; p < 3; p++, runBase += 2) {
            hipLaunchKernelGGL(yk_range1d_kernel<false>, dim3(nStrips), dim3(64), 0, c->stream, c->plane[0], c->plane[1], c->plane[2], c->strideElems,
                               c->fullW, c->h, c->covCh + (size_t)p * c->covChStride, c->mtW, c->tilesW, T8, c->r1Slots, c->r1Params, cT, cP, p, R1Direct{});
            hipLaunchKernelGGL(yk_u32_blocksum_kernel, dim3((unsigned)nb), dim3(1024), 0, c->stream, cT, T8, bT);
            hipLaunchKernelGGL(yk_u32_scanblocks_kernel, dim3(1), dim3(1024), 0, c->stream, bT, (int)nb, tot);
            hipLaunchKernelGGL(yk_u32_blocksum_kernel, dim3((unsigned)nb), dim3(1024), 0, c->stream, cP, T8, bP);
            hipLaunchKernelGGL(yk_u32_scanblocks_kernel, dim3(1), dim3(1024), 0, c->stream, bP, (int)nb, tot + 1);
            hipLaunchKernelGGL(yk_range1d_pack_kernel, dim3((unsigned)nb, 1), dim3(1024), 0, c->stream, cT, cP, bT, bP, tot, T8, c->r1Slots, c->r1Params, c->r1Pix, c->r1Type,
                               p, (const uint32_t*)runBase);
            hipLaunchKernelGGL(yk_r1_next_plane_kernel, dim3(1), dim3(64), 0, c->stream, runBase, (const uint32_t*)tot);
        }
        YK_HIP(c, hipGetLastError());
        uint32_t ends[8];
        YK_HIP(c, hipMemcpyAsync(ends, tot + 4, sizeof ends, hipMemcpyDeviceToHost, c->stream));
        YK_HIP(c, hipStreamSynchronize(c->stream));
        for (int p = 0; p < 3; p++) { c->r1EndTiles[p] = ends[2 * p + 2]; c->r1EndPix[p] = ends[2 * p + 3]; }
        t[0] = ends[6]; t[1] = ends[7];
    }
    c->r1Tiles = t[0]; c->r1PixCount = t[1]; c->r1Ready = true; c->r1TotalsPending = false;   // tile-planes coded and pixel bytes, all three planes
    return YK_OK;
}

static int yk_range1d_finish(yk_ctx* c) {
    if (!c->r1TotalsPending) return YK_OK;
    uint32_t t[2];
    YK_HIP(c, hipSetDevice(c->device));
    YK_HIP(c, hipMemcpyAsync(t, c->r1TotalsDev, sizeof t, hipMemcpyDeviceToHost, c->stream));
    YK_HIP(c, hipStreamSynchronize(c->stream));
    for (int p = 0; p < 3; p++) { c->r1EndTiles[p] = t[0] * (p + 1); c->r1EndPix[p] = t[1] * (p + 1); }
    c->r1Tiles = t[0] * 3; c->r1PixCount = t[1] * 3; c->r1TotalsPending = false;
    return YK_OK;
}

int yk_range1d_plane_ends(yk_ctx* c, size_t pixEnd[3], size_t typeEnd[3]) {
    if (!c) return YK_ERR_BAD_ARG;
    if (!c->r1Ready) return yk_fail(c, YK_ERR_STATE, "yk_range1d_encode first");
    { int rc = yk_range1d_finish(c); if (rc) return rc; }
    for (int p = 0; p < 3; p++) { if (pixEnd) pixEnd[p] = c->r1EndPix[p]; if (typeEnd) typeEnd[p] = (size_t)c->r1EndTiles[p] * 3; }
    return YK_OK;
}

int yk_range1d_streams_device(yk_ctx* c, const uint8_t** devPix, size_t* nPix, const uint8_t** devType, size_t* nType) {
    if (!c || !devPix || !nPix || !devType || !nType) return YK_ERR_BAD_ARG;
    if (!c->r1Ready) return yk_fail(c, YK_ERR_STATE, "yk_range1d_encode first");
    { int rc = yk_range1d_finish(c); if (rc) return rc; }
    *devPix = c->r1Pix; *nPix = (size_t)c->r1PixCount; *devType = c->r1Type; *nType = (size_t)c->r1Tiles * 3;
    return YK_OK;
}

int yk_range1d_streams(yk_ctx* c, uint8_t* hostPix, size_t capPix, size_t* nPix, uint8_t* hostType, size_t capType, size_t* nType) {
    if (!c) return YK_ERR_BAD_ARG;
    if (!c->r1Ready) return yk_fail(c, YK_ERR_STATE, "yk_range1d_encode first");
    { int rc = yk_range1d_finish(c); if (rc) return rc; }
    const size_t np = (size_t)c->r1PixCount, nt = (size_t)c->r1Tiles * 3;
    if (nPix) *nPix = np;
    if (nType) *nType = nt;
    YK_HIP(c, hipSetDevice(c->device));
    if (hostPix) { if (capPix < np) return yk_fail(c, YK_ERR_RANGE, "pixel stream buffer too small"); if (np) YK_HIP(c, hipMemcpyAsync(hostPix, c->r1Pix, np, hipMemcpyDeviceToHost, c->stream)); }
    if (hostType) { if (capType < nt) return yk_fail(c, YK_ERR_RANGE, "type stream buffer too small"); if (nt) YK_HIP(c, hipMemcpyAsync(hostType, c->r1Type, nt, hipMemcpyDeviceToHost, c->stream)); }
    YK_HIP(c, hipStreamSynchronize(c->stream));
    return YK_OK;
}

}  // extern "C"
