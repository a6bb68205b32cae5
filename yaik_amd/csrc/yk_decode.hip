// yk_decode.hip — gfx950 kernels for the per-tile decode loops behind YAIK_DecodeImage's chunk switch
// (decoder/YAIK_API.cpp:731-1303):
//
//   yk_decode_gradient   a16  DecompressGradient16x16/16x8/8x16/8x8/8x4/4x8/4x4   (decoder/YAIK_Gradient.cpp:28..1418)
//   yk_decode_1d         a17  Decompress1D x 3 planes                              (decoder/YAIK_3DTile.cpp:24-240)
//   yk_decode_mask       a18  Decompress1BitTiled                                  (decoder/YAIK_Mipmap.cpp:23-154)
//
// HBM layout mirrors YAIK_Instance (include/YAIK_private.h:26-54): planeR|G|B u8 in 8x8 tiles (tile-major, 64 B/tile,
// include/YAIK.h:205-224), the mapRGB corner lattice ((W/4+1) x (H/4+1) x 3 u8) and tile4x4Mask (1 bit per 4x4 cell).
// The reference walks each bitmap sequentially and pops "not yet seen" corners off the colour stream; on the GPU the
// stream offset of every tile is a prefix sum over the bitmap words of the corners each tile is the FIRST to touch
// (first toucher by scan position = atomicMin of bitIndex<<2|corner), then all tiles of the pass render in parallel.
// Passes are launched in call order, so a later, overlapping tile overwrites an earlier one exactly like the reference.
#include "yk_common.h"
#include "yk_device.h"

struct DPassGeo { int sx, sy, bigX, bigY, bitCount, xBB, tilesPerRow; };

__host__ __device__ static inline DPassGeo yk_dpass_geo(int sx, int sy, int w) {
    DPassGeo g; g.sx = sx; g.sy = sy;
    g.bigX = sx == 2 ? 32 : 64; g.bigY = sy == 2 ? 32 : 64;            // getSwizzleSize, include/YAIK_private.h:212-276
    g.tilesPerRow = g.bigX >> sx; g.bitCount = g.tilesPerRow * (g.bigY >> sy);
    g.xBB = (w + g.bigX - 1) / g.bigX;
    return g;
}

__device__ __forceinline__ void yk_dtile_from_bit(const DPassGeo& g, uint32_t pos, int& x, int& y) {
    const uint32_t blk = pos / g.bitCount, t = pos % g.bitCount;
    x = (int)(blk % g.xBB) * g.bigX + (int)(t % g.tilesPerRow) * (1 << g.sx);
    y = (int)(blk / g.xBB) * g.bigY + (int)(t / g.tilesPerRow) * (1 << g.sy);
}

// phase 1: first toucher of every not-yet-loaded lattice point.  One thread per tile slot (bit) of the bitmap.
__global__ __launch_bounds__(256) void yk_dec_owner_kernel(const uint32_t* __restrict__ bitmap, size_t nBits, DPassGeo g, int w, int h, int latW,
                                                           const uint8_t* __restrict__ loaded, uint32_t* __restrict__ owner) {
    const size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= nBits || !((bitmap[pos >> 5] >> (pos & 31)) & 1u)) return;
    const int dx = 1 << (g.sx - 2), dy = 1 << (g.sy - 2);
    int x, y; yk_dtile_from_bit(g, (uint32_t)pos, x, y);
    if (x + (1 << g.sx) > w || y + (1 << g.sy) > h) return;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const size_t li = (size_t)((y >> 2) + ((k & 2) ? dy : 0)) * latW + (x >> 2) + ((k & 1) ? dx : 0);
        if (!(loaded[li] & 1)) atomicMin(&owner[li], ((uint32_t)pos << 2) | (uint32_t)k);      // mapRGBMask plane 0 (hasRGB)
    }
}

// phase 2 (EMIT=false): corners owned per workgroup of 1024 tile slots; phase 3 (EMIT=true): pop the colours into the
// lattice (:97-136) at the offsets an exclusive scan over the owned-corner counts gives.
template <bool EMIT>
__global__ __launch_bounds__(1024) void yk_dec_corner_kernel(const uint32_t* __restrict__ bitmap, size_t nBits, DPassGeo g, int w, int h, int latW,
                                                             uint8_t* __restrict__ loaded, const uint32_t* __restrict__ owner, uint32_t* __restrict__ blockSums,
                                                             const uint8_t* __restrict__ rgb, size_t rgbBytes, uint8_t* __restrict__ mapRGB) {
    __shared__ uint32_t s_tmp[32];
    const size_t pos = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const int dx = 1 << (g.sx - 2), dy = 1 << (g.sy - 2);
    bool set = pos < nBits && ((bitmap[pos >> 5] >> (pos & 31)) & 1u);
    int x = 0, y = 0;
    if (set) { yk_dtile_from_bit(g, (uint32_t)pos, x, y); set = !(x + (1 << g.sx) > w || y + (1 << g.sy) > h); }
    uint32_t own = 0;
    size_t li[4] = { 0, 0, 0, 0 };
    if (set) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            li[k] = (size_t)((y >> 2) + ((k & 2) ? dy : 0)) * latW + (x >> 2) + ((k & 1) ? dx : 0);
            own |= (owner[li[k]] == (((uint32_t)pos << 2) | (uint32_t)k)) ? (1u << k) : 0u;
        }
    }
    uint32_t tot;
    const uint32_t ex = yk_block_exscan((uint32_t)__popc(own), s_tmp, &tot);
    if (!EMIT) { if (threadIdx.x == 0) blockSums[blockIdx.x] = tot; return; }
    uint32_t off = (blockSums[blockIdx.x] + ex) * 3u;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (!((own >> k) & 1u)) continue;
#pragma unroll
        for (int c = 0; c < 3; c++) mapRGB[li[k] * 3 + c] = (off + c < rgbBytes) ? rgb[off + c] : 0;
        loaded[li[k]] |= 1;
        off += 3;
    }
}

// phase 4: integer bilinear fill with truncation (:160-188, :762-781, :1382-1401) + tile4x4Mask marking.
// One workgroup per bitmap word.  The word's tiles are listed in LDS and the (tile, channel, row, run) items are spread over all 256
// threads, so small tiles keep the workgroup as busy as large ones.  A thread produces a run of min(TX, 8) pixels of one row
// and channel: in the 8x8-tiled plane layout that run is contiguous and leaves as ONE 8-byte (or 4-byte) store.
#ifndef YK_DEC_ROWPAIR
#define YK_DEC_ROWPAIR 0                                                     // 1: two rows per item (16-byte stores) for tiles 8 or 16 pixels wide -- measured slower (105 against 97 us at 8192 x 8192)
#endif
template <int NW> struct DRenderLdsN { int xy[32 * NW][2]; uint8_t c[32 * NW][3][4]; };
typedef DRenderLdsN<1> DRenderLds;
// the tiles of NW bitmap words (word k: index wi[k], bits bits[k]) of one pass, rendered together by the whole workgroup (256 threads)
template <int NW>
__device__ __forceinline__ void yk_dec_render_words(DRenderLdsN<NW>& L, const uint32_t* wi, const uint32_t* bitsN, const DPassGeo& g, int w, int h, int latW,
                                                    const uint8_t* __restrict__ mapRGB, uint8_t* __restrict__ planes, size_t planeSize, int tileW,
                                                    uint32_t* __restrict__ tile4, int stride4) {
    const int TX = 1 << g.sx, TY = 1 << g.sy, dx = TX >> 2, dy = TY >> 2;
    const int lgGpr = g.sx > 3 ? g.sx - 3 : 0, lgPer = g.sy + lgGpr;            // runs per row, runs per channel (powers of two)
    const int GW = TX < 8 ? TX : 8, gpr = 1 << lgGpr, perCh = 1 << lgPer, nEl = 3 * perCh, sh = g.sx + g.sy;
    int nT = 0;
    {
        const int wk = threadIdx.x >> 5, bit = threadIdx.x & 31;
        int before = 0;
#pragma unroll
        for (int k = 0; k < NW; k++) { const int pc = __popc(bitsN[k]); before += (k < wk) ? pc : 0; nT += pc; }
        if (wk < NW && ((bitsN[wk] >> bit) & 1u)) {
            int x, y; yk_dtile_from_bit(g, wi[wk] * 32u + (uint32_t)bit, x, y);
            const int k = before + __popc(bitsN[wk] & ((1u << bit) - 1u));
            L.xy[k][0] = (x + TX > w || y + TY > h) ? -1 : x; L.xy[k][1] = y;
        }
    }
    __syncthreads();
    // the four corner colours of the word's tiles, once (every run of a tile used to fetch them again: 4 byte loads per 8-byte store)
    // one (unaligned) 4-byte load per corner = its three colours (the lattice array is allocated four bytes longer), not three byte loads
    for (int i = threadIdx.x; i < nT * 4; i += 256) {
        const int k = i >> 2, q = i & 3;
        const int x = L.xy[k][0], y = L.xy[k][1];
        if (x < 0) continue;
        const size_t li = (size_t)((y >> 2) + ((q & 2) ? dy : 0)) * latW + (x >> 2) + ((q & 1) ? dx : 0);
        uint32_t rgb; __builtin_memcpy(&rgb, mapRGB + li * 3, 4);
        L.c[k][0][q] = (uint8_t)rgb; L.c[k][1][q] = (uint8_t)(rgb >> 8); L.c[k][2][q] = (uint8_t)(rgb >> 16);
    }
    __syncthreads();
    if (GW == 8 && YK_DEC_ROWPAIR) {
        // tiles at least 8 pixels wide: an item = two rows of an 8-pixel run = 16 contiguous bytes of the 8x8-tiled plane (row pitch 8 inside a tile)
        const int nEl2 = nEl >> 1, lgPer2 = lgPer - 1;
        for (int item = threadIdx.x; item < nT * nEl2; item += 256) {
            const int ck = item >> lgPer2, k = ck / 3, c = ck - 3 * k;           // tile of the list, channel
            const int x = L.xy[k][0], y = L.xy[k][1];
            if (x < 0) continue;
            const int r = item & ((perCh >> 1) - 1), ty = (r >> lgGpr) * 2, gx = (r & (gpr - 1)) * 8;
            const uint32_t c4 = *reinterpret_cast<const uint32_t*>(&L.c[k][c][0]);
            const int TL = c4 & 255, TR = (c4 >> 8) & 255, BL = c4 >> 16 & 255, BR = c4 >> 24;
            const int xx = x + gx, yy = y + ty;
            uint32_t o4[4];
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                const int Lf = TL * (TY - ty - rr) + BL * (ty + rr), R = TR * (TY - ty - rr) + BR * (ty + rr);
                int v = Lf * (TX - gx) + R * gx;                                  // numerator at the first pixel of the run, + (R - L) per pixel
                uint32_t lo = 0, hi = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) { lo |= (uint32_t)((v >> sh) & 255) << (8 * i); v += R - Lf; }
#pragma unroll
                for (int i = 0; i < 4; i++) { hi |= (uint32_t)((v >> sh) & 255) << (8 * i); v += R - Lf; }
                o4[rr * 2] = lo; o4[rr * 2 + 1] = hi;
            }
            uint8_t* o = planes + (size_t)c * planeSize + ((size_t)(yy >> 3) * tileW + (xx >> 3)) * 64 + (yy & 7) * 8;
            *reinterpret_cast<uint4*>(o) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
        }
    } else {
        for (int item = threadIdx.x; item < nT * nEl; item += 256) {
            const int ck = item >> lgPer, k = ck / 3, c = ck - 3 * k;           // tile of the list, channel
            const int x = L.xy[k][0], y = L.xy[k][1];
            if (x < 0) continue;
            const int r = item & (perCh - 1), ty = r >> lgGpr, gx = (r & (gpr - 1)) * GW;
            const uint32_t c4 = *reinterpret_cast<const uint32_t*>(&L.c[k][c][0]);
            const int TL = c4 & 255, TR = (c4 >> 8) & 255, BL = (c4 >> 16) & 255, BR = c4 >> 24;
            const int Lf = TL * (TY - ty) + BL * ty, R = TR * (TY - ty) + BR * ty;
            int v = Lf * (TX - gx) + R * gx;                                    // numerator at the first pixel of the run, + (R - L) per pixel
            const int xx = x + gx, yy = y + ty;
            uint8_t* o = planes + (size_t)c * planeSize + ((size_t)(yy >> 3) * tileW + (xx >> 3)) * 64 + (yy & 7) * 8 + (xx & 7);
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) { lo |= (uint32_t)((v >> sh) & 255) << (8 * i); v += R - Lf; }
            if (GW == 8) {
#pragma unroll
                for (int i = 0; i < 4; i++) { hi |= (uint32_t)((v >> sh) & 255) << (8 * i); v += R - Lf; }
                *reinterpret_cast<uint2*>(o) = make_uint2(lo, hi);
            } else *reinterpret_cast<uint32_t*>(o) = lo;
        }
    }
    // cell (cx,cy) -> byte (cx>>2) + (cy>>1)*stride4, bit ((cx>>1)&1)*4 + (cy&1)*2 + (cx&1)   (e.g. YAIK_Gradient.cpp:951-953).  A tile's
    // cells lie in at most two mask bytes (one per pair of cell rows; tiles are at most 4 cells wide and aligned to their size): one
    // atomic per byte with all of the tile's bits in it (one per CELL was 16 device-scope atomics per 16x16 tile: most of this kernel)
    for (int item = threadIdx.x; item < nT * 2; item += 256) {
        const int k = item >> 1, j = item & 1;
        const int x = L.xy[k][0], y = L.xy[k][1];
        if (x < 0) continue;
        const int cx0 = x >> 2, cy0 = y >> 2, br = (cy0 >> 1) + j;                // byte row
        if (br > ((cy0 + dy - 1) >> 1)) continue;
        uint32_t m = 0;
        for (int cy = max(cy0, br * 2); cy < min(cy0 + dy, br * 2 + 2); cy++)
            for (int cx = cx0; cx < cx0 + dx; cx++) m |= 1u << ((((cx >> 1) & 1) << 2) | ((cy & 1) << 1) | (cx & 1));
        const size_t byteIdx = (size_t)(cx0 >> 2) + (size_t)br * stride4;
        atomicOr(&tile4[byteIdx >> 2], m << (8 * (byteIdx & 3)));      // (collecting a 64x64 block's marks in LDS and merging them with eight atomics was slower: 105 against 97 us)
    }
}
__device__ __forceinline__ void yk_dec_render_word(DRenderLds& L, const size_t wi, const uint32_t bits, const DPassGeo& g, int w, int h, int latW,
                                                   const uint8_t* __restrict__ mapRGB, uint8_t* __restrict__ planes, size_t planeSize, int tileW,
                                                   uint32_t* __restrict__ tile4, int stride4) {
    const uint32_t wi1[1] = { (uint32_t)wi }, b1[1] = { bits };
    yk_dec_render_words<1>(L, wi1, b1, g, w, h, latW, mapRGB, planes, planeSize, tileW, tile4, stride4);
}

__global__ __launch_bounds__(256) void yk_dec_render_kernel(const uint32_t* __restrict__ bitmap, size_t nWords, DPassGeo g, int w, int h, int latW,
                                                            const uint8_t* __restrict__ mapRGB, uint8_t* __restrict__ planes, size_t planeSize, int tileW,
                                                            uint32_t* __restrict__ tile4, int stride4) {
    __shared__ DRenderLds L;
    const size_t wi = blockIdx.x;
    const uint32_t bits = bitmap[wi];
    if (!bits) return;
    yk_dec_render_word(L, wi, bits, g, w, h, latW, mapRGB, planes, planeSize, tileW, tile4, stride4);
}

// ---- all gradient chunks of a file in one call (yk_decode_gradient_all_device): the passes share every launch except the render --------------
// What the per-pass sequence above does pass after pass (first toucher among the lattice points no EARLIER pass loaded, count, scan, pop the
// colours) is one first-toucher problem over all passes when the key carries the pass: owner = min(pass << 27 | bitIndex << 2 | corner), the
// layout of yk_corners.hip on the encoder side.  Bitmap bytes (8 tile slots) of all passes are laid end to end; 1024 bytes per workgroup.
struct DecPlan {
    uint32_t byteStart[8], blockStart[8], wordStart[8];     // first bitmap byte / 1024-byte block / list word of pass p ([n..7] = totals)
    uint32_t nBytes[7], rgbBytes[7];
    const uint8_t* bm[7];
    const uint8_t* rgb[7];
    int sx[7], sy[7];
    int n;
};
__device__ __forceinline__ int yk_dplan_find(const uint32_t (&start)[8], uint32_t i) {
    int p = 0;
#pragma unroll
    for (int k = 1; k < 7; k++) p += (i >= start[k]) ? 1 : 0;
    return p;
}
// tiles of a bitmap byte: the 8 slots lie in one swizzle block (every block holds a multiple of 8 tiles); tiles per row / block are powers of two
struct DByteGeo { int bx0, by0, tprShift, tprMask; uint32_t t0; };
__device__ __forceinline__ DByteGeo yk_dbyte_geo(const DPassGeo& g, uint32_t bi) {
    DByteGeo b;
    const uint32_t pos0 = bi * 8u, blk = pos0 / (uint32_t)g.bitCount;
    b.t0 = pos0 % (uint32_t)g.bitCount;
    b.bx0 = (int)(blk % (uint32_t)g.xBB) * g.bigX; b.by0 = (int)(blk / (uint32_t)g.xBB) * g.bigY;
    b.tprShift = __ffs(g.tilesPerRow) - 1; b.tprMask = g.tilesPerRow - 1;
    return b;
}

__global__ __launch_bounds__(256) void yk_decall_owner_kernel(const DecPlan pl, int w, int h, int latW, const uint8_t* __restrict__ loaded, uint32_t* __restrict__ owner) {
    const uint32_t gi = blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= pl.byteStart[7]) return;
    const int pass = yk_dplan_find(pl.byteStart, gi);
    const uint32_t bi = gi - pl.byteStart[pass];
    const uint32_t byte = pl.bm[pass][bi];
    if (!byte) return;
    const DPassGeo g = yk_dpass_geo(pl.sx[pass], pl.sy[pass], w);
    const DByteGeo b = yk_dbyte_geo(g, bi);
    const int dx = 1 << (g.sx - 2), dy = 1 << (g.sy - 2);
    // tiles of the byte that lie inside the image (the reference skips the others, :58-60)
    uint32_t live = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t t = b.t0 + (uint32_t)k;
        const int x = b.bx0 + (int)((t & (uint32_t)b.tprMask) << g.sx), y = b.by0 + (int)((t >> b.tprShift) << g.sy);
        if (((byte >> k) & 1u) && !(x + (1 << g.sx) > w || y + (1 << g.sy) > h)) live |= 1u << k;
    }
    const bool two = g.tilesPerRow == 4;                                     // the byte is one row of 8 tiles, or two rows of 4
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (!((live >> k) & 1u)) continue;
        const uint32_t t = b.t0 + (uint32_t)k;
        const int x = b.bx0 + (int)((t & (uint32_t)b.tprMask) << g.sx), y = b.by0 + (int)((t >> b.tprShift) << g.sy);
        const uint32_t key = ((uint32_t)pass << 27) | ((bi * 8u + (uint32_t)k) << 2);
        // a corner an EARLIER live tile of this same byte also touches belongs to that tile (smaller scan position): no atomic for it
        // (left, upper, upper-left, upper-right neighbours; yk_corners.hip measured 32 -> 15 atomics for a dense group of 16x16 tiles)
        const int col = (int)(t & (uint32_t)b.tprMask);
        const bool left = k >= 1 && col != 0 && ((live >> (k - 1)) & 1u);
        const bool up = two && k >= 4 && ((live >> (k - 4)) & 1u);
        const bool upLeft = two && k >= 5 && col != 0 && ((live >> (k - 5)) & 1u);
        const bool upRight = two && k >= 4 && col != 3 && ((live >> (k - 3)) & 1u);
        const bool skip[4] = { left || up || upLeft, up || upRight, left, false };
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (skip[q]) continue;
            const size_t li = (size_t)((y >> 2) + ((q & 2) ? dy : 0)) * latW + (x >> 2) + ((q & 1) ? dx : 0);
            if (!(loaded[li] & 1)) atomicMin(&owner[li], key | (uint32_t)q);
        }
    }
}

// COUNT: corners owned per thread (a byte, kept for the emit launch) and per workgroup, non-empty bitmap words per workgroup.
// EMIT: the owners pop their colours off the pass's stream (offset = scanned corners before them x 3) into the lattice; the workgroup's
// non-empty words are appended to the pass's render list.
template <bool EMIT>
__global__ __launch_bounds__(1024) void yk_decall_stream_kernel(const DecPlan pl, int w, int h, int latW, uint8_t* __restrict__ loaded, const uint32_t* __restrict__ owner,
                                                                uint32_t* __restrict__ blockSums, uint32_t* __restrict__ blockWords, uint32_t* __restrict__ perThread,
                                                                uint8_t* __restrict__ mapRGB, uint32_t* __restrict__ wordList, uint32_t factor) {
    __shared__ uint32_t s_tmp[32];
    const int pass = yk_dplan_find(pl.blockStart, blockIdx.x);
    const uint32_t bi = (blockIdx.x - pl.blockStart[pass]) * 1024u + threadIdx.x;
    const uint32_t nBytes = pl.nBytes[pass];
    const size_t ti = (size_t)pl.byteStart[pass] + bi;
    const uint8_t* const bm = pl.bm[pass];
    const uint32_t byte = bi < nBytes ? bm[bi] : 0u;
    // a word of the bitmap is 4 consecutive bytes: its first thread speaks for it
    bool wordSet = false;
    if ((threadIdx.x & 3) == 0 && bi < nBytes) {
        wordSet = byte != 0u;
#pragma unroll
        for (int j = 1; j < 4; j++) wordSet |= (bi + j < nBytes) && bm[bi + j] != 0;
    }
    const DPassGeo g = yk_dpass_geo(pl.sx[pass], pl.sy[pass], w);
    const int dx = 1 << (g.sx - 2), dy = 1 << (g.sy - 2);
    // COUNT leaves the thread's ownership bits (4 per tile slot) for EMIT, which then neither repeats the 32 owner look-ups nor waits for them
    uint32_t cnt = 0, ownBits = 0;
    if (EMIT) { ownBits = bi < nBytes ? perThread[ti] : 0u; cnt = (uint32_t)__popc(ownBits); }
    uint32_t own[8];
    int tx[8], ty[8];
    if ((!EMIT && byte) || cnt) {
        const DByteGeo b = yk_dbyte_geo(g, bi);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t t = b.t0 + (uint32_t)k;
            tx[k] = b.bx0 + (int)((t & (uint32_t)b.tprMask) << g.sx); ty[k] = b.by0 + (int)((t >> b.tprShift) << g.sy);
        }
        if (EMIT) {
#pragma unroll
            for (int k = 0; k < 8; k++) own[k] = (ownBits >> (4 * k)) & 15u;
        } else {
            uint32_t o[8][4];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const bool set = ((byte >> k) & 1u) && !(tx[k] + (1 << g.sx) > w || ty[k] + (1 << g.sy) > h);
                const size_t l0 = set ? (size_t)(ty[k] >> 2) * latW + (tx[k] >> 2) : 0;
                const size_t ddx = set ? dx : 0, ddy = set ? (size_t)dy * latW : 0;
                o[k][0] = owner[l0]; o[k][1] = owner[l0 + ddx]; o[k][2] = owner[l0 + ddy]; o[k][3] = owner[l0 + ddy + ddx];
                const uint32_t key = ((uint32_t)pass << 27) | ((bi * 8u + (uint32_t)k) << 2);
                own[k] = set ? ((o[k][0] == (key | 0u) ? 1u : 0u) | (o[k][1] == (key | 1u) ? 2u : 0u) | (o[k][2] == (key | 2u) ? 4u : 0u) | (o[k][3] == (key | 3u) ? 8u : 0u)) : 0u;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) { own[k] = 0; tx[k] = 0; ty[k] = 0; }
    }
    if (!EMIT) {
#pragma unroll
        for (int k = 0; k < 8; k++) { cnt += (uint32_t)__popc(own[k]); ownBits |= own[k] << (4 * k); }
        uint32_t tot, totW;
        yk_block_exscan(cnt, s_tmp, &tot);
        yk_block_exscan(wordSet ? 1u : 0u, s_tmp, &totW);
        if (bi < nBytes) perThread[ti] = ownBits;
        if (threadIdx.x == 0) { blockSums[blockIdx.x] = tot; blockWords[blockIdx.x] = totW; }
        return;
    }
    uint32_t tot, totW;
    const uint32_t ex = yk_block_exscan(cnt, s_tmp, &tot);
    const uint32_t exW = yk_block_exscan(wordSet ? 1u : 0u, s_tmp, &totW);
    if (wordSet) wordList[pl.wordStart[pass] + (blockWords[blockIdx.x] - blockWords[pl.blockStart[pass]]) + exW] = bi >> 2;
    constexpr uint32_t kCap = 8192;                                          // colours listed per workgroup
    __shared__ uint32_t s_li[kCap];
    const uint32_t blockOff = (blockSums[blockIdx.x] - blockSums[pl.blockStart[pass]]) * 3u;
    const uint8_t* const rgb = pl.rgb[pass];
    const uint32_t rgbBytes = pl.rgbBytes[pass];
    auto colour = [&](const uint32_t li, const uint32_t j) {                 // the j-th colour of the workgroup goes to lattice point li (:97-136)
        const uint32_t off = blockOff + j * 3u;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            uint32_t v = (off + ch < rgbBytes) ? rgb[off + ch] : 0u;
            if (factor) v = (v * factor) >> 16;                              // PaletteFullRangeRemapping (YAIK_GenericFunctions.cpp:128-137)
            mapRGB[(size_t)li * 3 + ch] = (uint8_t)v;
        }
        loaded[li] |= 1;
    };
    const bool viaLds = tot <= kCap;
    if (cnt) {
        uint32_t j = ex;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (!own[k]) continue;
#pragma unroll
            for (int c4 = 0; c4 < 4; c4++) {
                if (!((own[k] >> c4) & 1u)) continue;
                const uint32_t li = (uint32_t)((ty[k] >> 2) + ((c4 & 2) ? dy : 0)) * (uint32_t)latW + (uint32_t)((tx[k] >> 2) + ((c4 & 1) ? dx : 0));
                if (viaLds) s_li[j] = li; else colour(li, j);
                j++;
            }
        }
    }
    if (!viaLds) return;
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < tot; j += 1024) colour(s_li[j], j);
}

// exclusive prefixes of both per-block arrays in place (one workgroup: thread t owns a run of consecutive blocks), words listed per pass
__global__ __launch_bounds__(1024) void yk_decall_scan_kernel(uint32_t* __restrict__ blockSums, uint32_t* __restrict__ blockWords, const DecPlan pl, uint32_t* __restrict__ passWords) {
    __shared__ uint32_t s_tmp[32];
    __shared__ uint32_t s_totalW;
    const uint32_t n = pl.blockStart[7], per = (n + 1023) / 1024;
    const uint32_t a = min(threadIdx.x * per, n), b = min(a + per, n);
    uint32_t sum = 0, sumW = 0;
    for (uint32_t i = a; i < b; i++) { sum += blockSums[i]; sumW += blockWords[i]; }
    uint32_t tot, totW;
    uint32_t run = yk_block_exscan(sum, s_tmp, &tot);
    uint32_t runW = yk_block_exscan(sumW, s_tmp, &totW);
    for (uint32_t i = a; i < b; i++) {
        const uint32_t v = blockSums[i], vw = blockWords[i];
        blockSums[i] = run; run += v; blockWords[i] = runW; runW += vw;
    }
    if (threadIdx.x == 0) s_totalW = totW;
    __syncthreads();
    if (threadIdx.x < 7) {
        const uint32_t lo = pl.blockStart[threadIdx.x] < n ? blockWords[pl.blockStart[threadIdx.x]] : s_totalW;
        const uint32_t hi = pl.blockStart[threadIdx.x + 1] < n ? blockWords[pl.blockStart[threadIdx.x + 1]] : s_totalW;
        passWords[threadIdx.x] = hi - lo;
    }
}

// render of one pass from its list of non-empty bitmap words (a launch over every word of a sparse map is mostly workgroups that start and
// leave: 336 k of them over the seven passes of an 8192 x 8192 frame)
__global__ __launch_bounds__(256) void yk_decall_render_kernel(const uint8_t* __restrict__ bm, uint32_t nBytes, const uint32_t* __restrict__ wordList, const uint32_t* __restrict__ nListed,
                                                               DPassGeo g, int w, int h, int latW, const uint8_t* __restrict__ mapRGB, uint8_t* __restrict__ planes,
                                                               size_t planeSize, int tileW, uint32_t* __restrict__ tile4, int stride4) {
    __shared__ DRenderLds L;
    const uint32_t n = *nListed;
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        const uint32_t wi = wordList[i], b0 = wi * 4u;
        uint32_t bits = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) if (b0 + j < nBytes) bits |= (uint32_t)bm[b0 + j] << (8 * j);
        yk_dec_render_word(L, wi, bits, g, w, h, latW, mapRGB, planes, planeSize, tileW, tile4, stride4);
        __syncthreads();
    }
}

// ONE render launch for all passes: a workgroup owns a 64x64 block of the image and walks the passes in call order, rendering the tiles of the
// bitmap words that cover its block (every tile shape's swizzle blocks are 64 or 32 pixels wide and high: at most 20.5 words per block and frame).
// A later, overlapping tile still overwrites an earlier one like the reference (same workgroup, passes in order, stores of a pass acknowledged
// before the next one starts); no lists of non-empty words are needed (an empty block costs its workgroup seven small loads), and seven dependent
// launches of 16 us each (0.11 ms of the 0.33 ms decode of an 8192 x 8192 frame) become one.
__global__ __launch_bounds__(256) void yk_decall_render_blocks_kernel(const DecPlan pl, int w, int h, int latW, const uint8_t* __restrict__ mapRGB, uint8_t* __restrict__ planes,
                                                                      size_t planeSize, int tileW, uint32_t* __restrict__ tile4, int stride4) {
    __shared__ DRenderLdsN<8> L;
    __shared__ uint32_t s_word[7][8], s_idx[7][8];
    const int xB64 = (w + 63) >> 6;
    const int bx64 = (int)(blockIdx.x % (unsigned)xB64), by64 = (int)(blockIdx.x / (unsigned)xB64);
    // the (up to 8) words of every pass that cover this block, fetched together: thread = pass * 8 + slot
    if (threadIdx.x < 56) {
        const int p = threadIdx.x >> 3, k = threadIdx.x & 7;
        uint32_t bits = 0, wi = 0;
        if (p < pl.n) {
            const DPassGeo g = yk_dpass_geo(pl.sx[p], pl.sy[p], w);
            const int nbx = 64 / g.bigX, nby = 64 / g.bigY, wpb = g.bitCount >= 32 ? g.bitCount >> 5 : 1;      // swizzle blocks per 64x64 block, words per swizzle block
            const int sb = k / wpb, wk = k - sb * wpb;
            if (sb < nbx * nby) {
                const int sbx = bx64 * nbx + (sb % nbx), sby = by64 * nby + (sb / nbx), yBB = (h + g.bigY - 1) / g.bigY;
                if (sbx < g.xBB && sby < yBB) {
                    const uint32_t bit0 = (uint32_t)(sby * g.xBB + sbx) * (uint32_t)g.bitCount;
                    wi = (bit0 >> 5) + (uint32_t)wk;
                    const uint32_t b0 = wi * 4u;
#pragma unroll
                    for (int j = 0; j < 4; j++) if (b0 + j < pl.nBytes[p]) bits |= (uint32_t)pl.bm[p][b0 + j] << (8 * j);
                    if (g.bitCount < 32) bits &= 0xFFFFu << (bit0 & 31u);        // 16x16: two blocks share a word
                }
            }
        }
        s_word[p][k] = bits; s_idx[p][k] = wi;
    }
    __syncthreads();
    bool wrote = false;
    for (int p = 0; p < pl.n; p++) {
        uint32_t bits8[8], idx8[8], any = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) { bits8[k] = s_word[p][k]; idx8[k] = s_idx[p][k]; any |= bits8[k]; }
        if (!any) continue;                                                       // uniform over the workgroup
        if (wrote) { __threadfence_block(); __syncthreads(); }                   // the previous pass's stores are out and its LDS image is free
        const DPassGeo g = yk_dpass_geo(pl.sx[p], pl.sy[p], w);
        yk_dec_render_words<8>(L, idx8, bits8, g, w, h, latW, mapRGB, planes, planeSize, tileW, tile4, stride4);
        wrote = true;
    }
}

// ---- partial planes: DecompressGradient4x4R / G / B / RG / GB / RB (decoder/YAIK_Gradient.cpp:1208-1226, :1420-2732) ------------
// `loaded` holds one bit per plane and lattice point (the three planes of mapRGBMask); the masks are split first (UpdateTileAndRGBMask).
__global__ void yk_dec_split_kernel(uint8_t* __restrict__ loaded, size_t lat, uint8_t* __restrict__ tile4, size_t tile4Size) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < lat) loaded[i] = (loaded[i] & 1) ? 7 : 0;                           // planes 1, 2 <- plane 0 (YAIK_API.cpp:536-537)
    if (i < tile4Size) { const uint8_t v = tile4[i]; tile4[tile4Size + i] = v; tile4[2 * tile4Size + i] = v; }     // :539-540
}
// 4x4 tiles: 32x32 swizzle blocks of 64 tiles, bit t -> tile (t % 8, t / 8)
__device__ __forceinline__ bool yk_d44_tile(const uint32_t* bitmap, size_t nBits, size_t pos, int w, int h, int& x, int& y) {
    if (pos >= nBits || !((bitmap[pos >> 5] >> (pos & 31)) & 1u)) return false;
    const int xBB = (w + 31) / 32;
    const uint32_t blk = (uint32_t)(pos >> 6), t = (uint32_t)(pos & 63);
    x = (int)(blk % xBB) * 32 + (int)(t & 7) * 4; y = (int)(blk / xBB) * 32 + (int)(t >> 3) * 4;
    return x < w && y < h;
}
__global__ __launch_bounds__(256) void yk_dec44p_owner_kernel(const uint32_t* __restrict__ bitmap, size_t nBits, int w, int h, int latW, int planeBit,
                                                              const uint8_t* __restrict__ loaded, uint32_t* __restrict__ owner) {
    const size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    int x, y;
    if (!yk_d44_tile(bitmap, nBits, pos, w, h, x, y)) return;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const size_t li = (size_t)((y >> 2) + (k >> 1)) * latW + (x >> 2) + (k & 1);
        if ((~loaded[li]) & planeBit) atomicMin(&owner[li], ((uint32_t)pos << 2) | (uint32_t)k);      // some present plane still lacks this point
    }
}
// EMIT=false: bytes popped per workgroup of 1024 tile slots; EMIT=true: pop them (per corner TL,TR,BL,BR, per present plane in R,G,B
// order, only planes that lack the point, e.g. :1516-1580), then mark the point for the present planes
template <bool EMIT>
__global__ __launch_bounds__(1024) void yk_dec44p_corner_kernel(const uint32_t* __restrict__ bitmap, size_t nBits, int w, int h, int latW, int planeBit,
                                                                uint8_t* __restrict__ loaded, const uint32_t* __restrict__ owner, uint32_t* __restrict__ blockSums,
                                                                const uint8_t* __restrict__ rgb, size_t rgbBytes, uint8_t* __restrict__ mapRGB) {
    __shared__ uint32_t s_tmp[32];
    const size_t pos = (size_t)blockIdx.x * 1024 + threadIdx.x;
    int x = 0, y = 0;
    const bool set = yk_d44_tile(bitmap, nBits, pos, w, h, x, y);
    uint32_t need[4] = { 0, 0, 0, 0 }, bytes = 0;
    size_t li[4] = { 0, 0, 0, 0 };
    if (set) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            li[k] = (size_t)((y >> 2) + (k >> 1)) * latW + (x >> 2) + (k & 1);
            if (owner[li[k]] == (((uint32_t)pos << 2) | (uint32_t)k)) { need[k] = (uint32_t)((~loaded[li[k]]) & planeBit); bytes += (uint32_t)__popc(need[k]); }
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
        for (int c = 0; c < 3; c++) if ((need[k] >> c) & 1u) { mapRGB[li[k] * 3 + c] = off < rgbBytes ? rgb[off] : 0; off++; }
    }
}
__global__ __launch_bounds__(256) void yk_dec44p_mark_kernel(const uint32_t* __restrict__ owner, size_t lat, int planeBit, uint8_t* __restrict__ loaded) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < lat && owner[i] != 0xFFFFFFFFu) loaded[i] |= (uint8_t)planeBit;    // every touched point now has the present planes
}
// fill (truncating integer bilinear, e.g. :1617-1640) + tile4x4Mask marking.  consistentMarks = 0 reproduces what the reference's loops
// do to the mask: R / G / B never mark, GB / RB put the B marks at tile4x4Mask + (tile4x4MaskSize >> 1) (:1678, :1924).
__global__ __launch_bounds__(256) void yk_dec44p_render_kernel(const uint32_t* __restrict__ bitmap, size_t nBits, int w, int h, int latW, int planeBit, int consistentMarks,
                                                               const uint8_t* __restrict__ mapRGB, uint8_t* __restrict__ planes, size_t planeSize, int tileW,
                                                               uint32_t* __restrict__ tile4, size_t tile4Size, int stride4) {
    const size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    int x, y;
    if (!yk_d44_tile(bitmap, nBits, pos, w, h, x, y)) return;
    const size_t l0 = (size_t)(y >> 2) * latW + (x >> 2);
    for (int c = 0; c < 3; c++) {
        if (!((planeBit >> c) & 1)) continue;
        const int TL = mapRGB[l0 * 3 + c], TR = mapRGB[(l0 + 1) * 3 + c], BL = mapRGB[(l0 + latW) * 3 + c], BR = mapRGB[(l0 + latW + 1) * 3 + c];
        uint8_t* o = planes + (size_t)c * planeSize + ((size_t)(y >> 3) * tileW + (x >> 3)) * 64 + (y & 7) * 8 + (x & 7);
#pragma unroll
        for (int ty = 0; ty < 4; ty++) {
            const int L = TL * (4 - ty) + BL * ty, R = TR * (4 - ty) + BR * ty;
            uint32_t v4 = 0;
#pragma unroll
            for (int tx = 0; tx < 4; tx++) v4 |= (uint32_t)(((L * (4 - tx) + R * tx) >> 4) & 255) << (8 * tx);
            *reinterpret_cast<uint32_t*>(o + ty * 8) = v4;
        }
        size_t base;
        if (consistentMarks) base = tile4Size * c;
        else {
            if (planeBit == 1 || planeBit == 2 || planeBit == 4) continue;
            base = (c == 2) ? (tile4Size >> 1) : tile4Size * c;
        }
        const int cx = x >> 2, cy = y >> 2;
        const size_t byteIdx = base + (size_t)(cx >> 2) + (size_t)(cy >> 1) * stride4;
        const uint32_t bit = (uint32_t)((((cx >> 1) & 1) << 2) | ((cy & 1) << 1) | (cx & 1));
        atomicOr(&tile4[byteIdx >> 2], 1u << (bit + 8 * (byteIdx & 3)));
    }
}

// ---- 1-D range decode -------------------------------------------------------------------------------------------
// quadrant mask of tile i (bit set = quadrant already filled by a gradient tile, YAIK_3DTile.cpp:74-78); 0xF past the end
__device__ __forceinline__ int yk_d1_quads(const uint8_t* __restrict__ tile4, int stride4, int tilesW, size_t T8, size_t i) {
    if (i >= T8) return 0xF;
    const int tx = (int)(i % tilesW), ty = (int)(i / tilesW);
    const int q = tile4[(tx >> 1) + (size_t)ty * stride4];
    return (q >> ((tx & 1) ? 4 : 0)) & 0xF;
}
// coded tiles (low 11 bits) and pixel bytes (above) of a tile, packed so that one scan serves both: a block of 1024 tiles holds at
// most 1024 coded tiles and 65536 pixel bytes
__device__ __forceinline__ uint32_t yk_d1_packed_count(int q) { return (q != 0xF ? 1u : 0u) | ((16u * (4u - (uint32_t)__popc(q))) << 11); }

// per block of 1024 tiles: coded tiles and pixel bytes, from the mask alone
__global__ __launch_bounds__(1024) void yk_dec1d_count_kernel(const uint8_t* __restrict__ tile4, int stride4, int tilesW, size_t T8,
                                                              uint32_t* __restrict__ blockTiles, uint32_t* __restrict__ blockPix, uint32_t* __restrict__ offInBlk) {
    __shared__ uint32_t s_tmp[32];
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    uint32_t tot;
    const int q = yk_d1_quads(tile4, stride4, tilesW, T8, i);
    const uint32_t e = yk_block_exscan(yk_d1_packed_count(q), s_tmp, &tot);
    if (i < T8) offInBlk[i] = e | ((uint32_t)q << 28);                         // the tile's packed offsets inside its block (28 bits) + its quadrant mask
    if (threadIdx.x == 0) { blockTiles[blockIdx.x] = tot & 2047u; blockPix[blockIdx.x] = tot >> 11; }
}
// both block-sum arrays -> exclusive prefixes in place, totals[0] = coded tiles, totals[1] = pixel bytes (one launch)
__global__ __launch_bounds__(1024) void yk_dec1d_scan_kernel(uint32_t* __restrict__ blockTiles, uint32_t* __restrict__ blockPix, int nBlocks, uint32_t* __restrict__ totals) {
    __shared__ uint32_t s_tmp[32];
#pragma unroll 1
    for (int a = 0; a < 2; a++) {
        uint32_t* arr = a ? blockPix : blockTiles;
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

// FOUR lanes per tile, one lane per pair of rows of one half of the tile; a tile's offsets in the type and pixel streams come from the count
// kernel's per-tile scan (offInBlk) + the scanned block sums, so the kernel is a plain stream: no workgroup scan, no barrier, every lane's loads
// independent of every other lane's (the first form scanned 1024 tiles per workgroup of 1024 threads and then walked them in four rounds: two
// resident workgroups per CU, each a chain of dependent round trips).  The stream holds a half's rows as [left quadrant 4 B][right quadrant
// 4 B] per row, absent quadrants left out (:95-124): with both quadrants present a row pair is 16 contiguous, 16-byte aligned stream bytes AND
// 16 contiguous bytes of the 8x8-tiled plane (one load, one store per lane, a wave moves 1 KB per instruction); with one quadrant it is an
// 8-byte load and two 4-byte stores.
#ifndef YK_D1_TPL
#define YK_D1_TPL 4
#endif
__global__ __launch_bounds__(256) void yk_dec1d_kernel(const uint32_t* __restrict__ offInBlk, size_t T8,
                                                       const uint32_t* __restrict__ baseTiles, const uint32_t* __restrict__ basePix,
                                                       const uint32_t* __restrict__ totals /*[0]=tiles,[1]=pix*/, const uint8_t* __restrict__ type, size_t typeBytes,
                                                       const uint8_t* __restrict__ pix, size_t pixBytes, int invRange,
                                                       uint8_t* __restrict__ planes, size_t planeSize, int planeOverride, const uint32_t* __restrict__ runBase) {
    // planeOverride < 0: the three planes share one mask (no partial-plane pass ran): plane = blockIdx.y, its streams start at
    // plane * totals.  Otherwise one plane per launch with its own mask, streams start at runBase (tiles, pixels of the planes before it).
    const int p = planeOverride < 0 ? (int)blockIdx.y : planeOverride;
    const size_t baseT = planeOverride < 0 ? (size_t)p * totals[0] : (size_t)runBase[0], baseP = planeOverride < 0 ? (size_t)p * totals[1] : (size_t)runBase[1];
    const int j = threadIdx.x & 3, half = j >> 1, rp = j & 1;
    uint8_t* const plane = planes + (size_t)p * planeSize;
    // YK_D1_TPL tiles per lane, 64 tiles apart (a wave's accesses stay contiguous), in three phases so that a lane's round trips overlap instead of
    // following each other: all offset words, then all parameter bytes and stream bytes, then the arithmetic and the stores.  (Tile after tile
    // a wave made eight dependent round trips; with eight waves per SIMD and 48 waves per SIMD to run, those chains were the launch's duration.)
    uint32_t ow[YK_D1_TPL];
#pragma unroll
    for (int rep = 0; rep < YK_D1_TPL; rep++) {
        const size_t i = ((size_t)blockIdx.x * YK_D1_TPL + rep) * 64 + (threadIdx.x >> 2);
        ow[rep] = i < T8 ? offInBlk[i] : 0xF0000000u;                           // past the end: a tile with nothing to decode
    }
    int tb[YK_D1_TPL]; uint4 L[YK_D1_TPL]; bool coded[YK_D1_TPL], live[YK_D1_TPL];
#pragma unroll
    for (int rep = 0; rep < YK_D1_TPL; rep++) {
        const int q = (int)(ow[rep] >> 28);
        const int qh = (q >> (half * 2)) & 3;                                    // bit 0: left quadrant filled, bit 1: right
        // the 64 tiles of a workgroup's round lie in one scan block: its two bases are scalar loads
        const size_t blk = (((size_t)blockIdx.x * YK_D1_TPL + rep) * 64) >> 10;
        const bool inRange = (((size_t)blockIdx.x * YK_D1_TPL + rep) * 64) < T8;
        const uint32_t offT = (inRange ? baseTiles[blk] : 0u) + (ow[rep] & 2047u), offP = (inRange ? basePix[blk] : 0u) + ((ow[rep] >> 11) & 0x1FFFFu);
        const int nTop = 2 - (q & 1) - ((q >> 1) & 1);
        const size_t to = (baseT + offT) * 3;
        // the tile's three parameter bytes: one byte load per lane (lane j of the tile fetches byte min(j, 2)), handed round the quad by DPP below
        live[rep] = q != 0xF;                                                     // the tile has a quadrant nothing filled: it is written here, with zeros when its parameters are missing
        coded[rep] = live[rep] && to + 2 < typeBytes;
        tb[rep] = 0;
        if (coded[rep]) tb[rep] = type[to + (j < 2 ? j : 2)];
        L[rep] = make_uint4(0u, 0u, 0u, 0u);
        if (coded[rep] && qh == 0) {
            const size_t po = baseP + offP + (half ? 16 * nTop : 0) + (size_t)rp * 16;
            if (po + 16 <= pixBytes) L[rep] = *reinterpret_cast<const uint4*>(pix + po);
            else if (po < pixBytes) { uint32_t tmp[4] = { 0, 0, 0, 0 }; for (int k = 0; k < 4; k++) if (po + 4 * k + 3 < pixBytes) tmp[k] = *reinterpret_cast<const uint32_t*>(pix + po + 4 * k); L[rep] = make_uint4(tmp[0], tmp[1], tmp[2], tmp[3]); }
        } else if (coded[rep] && qh != 3) {
            const size_t po = baseP + offP + (half ? 16 * nTop : 0) + (size_t)rp * 8;
            if (po + 8 <= pixBytes) { const uint2 t2 = *reinterpret_cast<const uint2*>(pix + po); L[rep].x = t2.x; L[rep].y = t2.y; }
            else if (po + 3 < pixBytes) L[rep].x = *reinterpret_cast<const uint32_t*>(pix + po);
        }
    }
#pragma unroll
    for (int rep = 0; rep < YK_D1_TPL; rep++) {
        const size_t i = ((size_t)blockIdx.x * YK_D1_TPL + rep) * 64 + (threadIdx.x >> 2);
        const int q = (int)(ow[rep] >> 28);
        const int qh = (q >> (half * 2)) & 3;
        const int color0 = __builtin_amdgcn_update_dpp(0, tb[rep], 0x00, 0xF, 0xF, true), base = __builtin_amdgcn_update_dpp(0, tb[rep], 0x55, 0xF, 0xF, true),
                  delta = __builtin_amdgcn_update_dpp(0, tb[rep], 0xAA, 0xF, 0xF, true);
        // a tile the stream does not reach decodes as color0 = 0 with every index 0: zeros.  (The planes are not cleared per frame any more: what no
        // gradient / LUT tile covered is written here, whatever the stream holds: yk_decode_begin.)
        if (qh == 3 || !live[rep]) continue;
        const int delta2 = ((delta * invRange) >> 8) + 1;                       // :66, :86
        // v = L ? base + (((L - 1) * delta2) >> 16) : color0 (:113-124), four pixels of a dword at a time: for L >= 1 the value is byte 2 of
        // K + L * delta2 with K = (base << 16) - delta2 (delta2 < 2^21, L a byte: a 24-bit multiply with the byte selected by the instruction);
        // the bytes with L == 0 are found with the carry-free zero-byte mask and take color0.  4.5 operations per pixel instead of 8.
        const uint32_t Kc = ((uint32_t)base << 16) - (uint32_t)delta2, d2u = (uint32_t)delta2, c0x4 = (uint32_t)color0 * 0x01010101u;
        auto dec4 = [&](uint32_t L4) {
            uint32_t t0, t1, t2, t3;
            asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(t0) : "v"(L4), "v"(d2u));
            asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(t1) : "v"(L4), "v"(d2u));
            asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(t2) : "v"(L4), "v"(d2u));
            asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(t3) : "v"(L4), "v"(d2u));
            t0 += Kc; t1 += Kc; t2 += Kc; t3 += Kc;
            const uint32_t lo = __builtin_amdgcn_perm(t1, t0, 0x0C0C0602u), hi = __builtin_amdgcn_perm(t3, t2, 0x06020C0Cu);   // byte 2 of each
            const uint32_t dec = lo | hi;
            const uint32_t z = ~(((L4 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | L4 | 0x7F7F7F7Fu);                                       // 0x80 in every byte with L == 0
            const uint32_t m = (z >> 7) * 255u;
            return (dec & ~m) | (c0x4 & m);
        };
        uint8_t* const o = plane + i * 64 + (half * 4 + rp * 2) * 8;               // the lane's two rows: 16 contiguous bytes of the tile
        if (qh == 0) *reinterpret_cast<uint4*>(o) = make_uint4(dec4(L[rep].x), dec4(L[rep].y), dec4(L[rep].z), dec4(L[rep].w));
        else {
            const int side = qh & 1;                                             // left filled -> the present quadrant is the right one
            *reinterpret_cast<uint32_t*>(o + side * 4) = dec4(L[rep].x);
            *reinterpret_cast<uint32_t*>(o + 8 + side * 4) = dec4(L[rep].y);
        }
    }
}

__global__ void yk_dec_mask_kernel(const uint8_t* __restrict__ bits, int bw, int bh, unsigned long long* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bw * bh) return;
    const int x = i % bw, y = i / bw;
    const unsigned long long v = (bits[i >> 3] & (1 << (i & 7))) ? ~0ULL : 0ULL;     // YAIK_Mipmap.cpp:119-136
    unsigned long long* A = out + (size_t)y * 4 * bw + 2 * x;
    A[0] = v; A[1] = v; A[2 * bw] = v; A[2 * bw + 1] = v;
}

// a20: the default image builder (decoder/YAIK_DefaultCallback.cpp:24-191): 8x8-tiled planes -> interleaved RGB rows at
// outputImageStride.  With an alpha plane the reference never advances past the alpha byte (:53-60) and produces
// 3-byte-strided garbage; this kernel writes proper RGBA (4 B/pixel) instead, which is what YAIK.h documents.
__global__ __launch_bounds__(256) void yk_dec_detile_kernel(const uint8_t* __restrict__ planes, size_t planeSize, int tileW, int w, int h,
                                                            const uint8_t* __restrict__ alpha, int strideA, uint8_t* __restrict__ out, size_t stride) {
    // one thread = 4 pixels of a row (half a tile row): one 4-byte load per plane, 12 (RGB) or 16 (RGBA) output bytes
    const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;                               // w is a multiple of 16
    const size_t ti = ((size_t)(y >> 3) * tileW + (x >> 3)) * 64 + (y & 7) * 8 + (x & 7);
    const uint32_t r = *reinterpret_cast<const uint32_t*>(planes + ti), g = *reinterpret_cast<const uint32_t*>(planes + planeSize + ti),
                   b = *reinterpret_cast<const uint32_t*>(planes + 2 * planeSize + ti);
    if (alpha) {
        uint8_t* o = out + (size_t)y * stride + (size_t)x * 4;
        const uint8_t* a = alpha + (size_t)y * strideA + x;
        uint32_t px[4];
#pragma unroll
        for (int k = 0; k < 4; k++) px[k] = ((r >> (8 * k)) & 255u) | (((g >> (8 * k)) & 255u) << 8) | (((b >> (8 * k)) & 255u) << 16) | ((uint32_t)a[k] << 24);
        if ((reinterpret_cast<uintptr_t>(o) & 3) == 0) { uint32_t* o4 = reinterpret_cast<uint32_t*>(o); o4[0] = px[0]; o4[1] = px[1]; o4[2] = px[2]; o4[3] = px[3]; }
        else for (int k = 0; k < 16; k++) o[k] = (uint8_t)(px[k >> 2] >> (8 * (k & 3)));
    } else {
        uint8_t* o = out + (size_t)y * stride + (size_t)x * 3;
        const uint32_t r0 = r & 255u, r1 = (r >> 8) & 255u, r2 = (r >> 16) & 255u, r3 = r >> 24;
        const uint32_t g0 = g & 255u, g1 = (g >> 8) & 255u, g2 = (g >> 16) & 255u, g3 = g >> 24;
        const uint32_t b0 = b & 255u, b1 = (b >> 8) & 255u, b2 = (b >> 16) & 255u, b3 = b >> 24;
        const uint32_t w0 = r0 | (g0 << 8) | (b0 << 16) | (r1 << 24), w1 = g1 | (b1 << 8) | (r2 << 16) | (g2 << 24), w2 = b2 | (r3 << 8) | (g3 << 16) | (b3 << 24);
        if ((reinterpret_cast<uintptr_t>(o) & 3) == 0) { uint32_t* o4 = reinterpret_cast<uint32_t*>(o); o4[0] = w0; o4[1] = w1; o4[2] = w2; }
        else {
            const uint32_t ws[3] = { w0, w1, w2 };
            for (int k = 0; k < 12; k++) o[k] = (uint8_t)(ws[k >> 2] >> (8 * (k & 3)));
        }
    }
}

// The reference's RGBA branch as it executes (decoder/YAIK_DefaultCallback.cpp:45-62): the alpha store does not advance dst, so a row is
// w RGB triples followed by ONE alpha byte; the alpha row cursors advance w bytes per tile row instead of 8 * strideA (:36-39 vs
// :128-130), so row y ends on the byte alpha[strideA * (y & 7) + w * (y >> 3) + w - 1].  One thread per row writes that byte.
__global__ __launch_bounds__(256) void yk_dec_ref_alpha_kernel(const uint8_t* __restrict__ alpha, int strideA, size_t alphaBytes, int w, int h,
                                                               uint8_t* __restrict__ out, size_t pitch) {
    const int y = blockIdx.x * 256 + threadIdx.x;
    if (y >= h) return;
    const size_t src = (size_t)strideA * (y & 7) + (size_t)w * (y >> 3) + (size_t)(w - 1);
    out[(size_t)y * pitch + (size_t)w * 3] = src < alphaBytes ? alpha[src] : 0;
}

// ---- host side ---------------------------------------------------------------------------------------------------
static int yk_dec_scratch(yk_ctx* c, size_t bytes) {
    if (c->dScratchBytes >= bytes) return YK_OK;
    if (c->dScratch) { YK_HIP(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->dScratch); c->dScratch = nullptr; }
    YK_HIP(c, hipMalloc(&c->dScratch, bytes));
    c->dScratchBytes = bytes;
    return YK_OK;
}

extern "C" {

// zero every 4x4 cell tile4x4Mask does not mark (thread = 8x8 tile and plane; the mask is shared by the planes until a plane-subset pass splits it)
__global__ __launch_bounds__(256) void yk_dec_zero_unmarked_kernel(const uint8_t* __restrict__ tile4, size_t tile4Size, int split, int stride4, int tilesW, size_t T8,
                                                                   uint8_t* __restrict__ planes, size_t planeSize) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int p = blockIdx.y;
    if (i >= T8) return;
    const int tx = (int)(i % (size_t)tilesW), ty = (int)(i / (size_t)tilesW);
    const uint8_t m = tile4[(split ? (size_t)p * tile4Size : 0) + (size_t)(tx >> 1) + (size_t)ty * stride4];     // byte = 16x8 pixels: bit ((cx>>1)&1)*4 + (cy&1)*2 + (cx&1)
    const int sh = (tx & 1) * 4;
    uint8_t* o = planes + (size_t)p * planeSize + i * 64;
#pragma unroll
    for (int qy = 0; qy < 2; qy++)
#pragma unroll
        for (int qx = 0; qx < 2; qx++)
            if (!((m >> (sh + qy * 2 + qx)) & 1))
                for (int r = 0; r < 4; r++) *reinterpret_cast<uint32_t*>(o + (qy * 4 + r) * 8 + qx * 4) = 0u;
}
static int yk_dec_settle(yk_ctx* c) {
    if (!c->dPlanesStale) return YK_OK;
    const int w = c->dw, h = c->dh;
    const size_t T8 = (size_t)(w >> 3) * (h >> 3);
    hipLaunchKernelGGL(yk_dec_zero_unmarked_kernel, dim3((unsigned)((T8 + 255) / 256), 3), dim3(256), 0, c->stream, c->dTile4, c->dTile4Size, c->dSplit ? 1 : 0,
                       (w + 15) >> 4, w >> 3, T8, c->dPlanes, c->dPlaneSize);
    YK_HIP(c, hipGetLastError());
    c->dPlanesStale = false;
    return YK_OK;
}

int yk_decode_begin(yk_ctx* c, int w, int h) {
    if (!c) return YK_ERR_BAD_ARG;
    if (w < 16 || h < 16 || (w & 15) || (h & 15) || w > 32752 || h > 32752) return yk_fail(c, YK_ERR_BAD_ARG, "decode needs width/height multiples of 16");
    YK_HIP(c, hipSetDevice(c->device));
    const size_t lat = (size_t)(w / 4 + 1) * (h / 4 + 1);
    if (!c->dPlanes || c->dw != w || c->dh != h) {                             // a stream of images of one shape keeps its buffers: only the clears below
        YK_HIP(c, hipStreamSynchronize(c->stream));
        auto F = [](auto*& p) { if (p) { (void)hipFree((void*)p); p = nullptr; } };
        F(c->dPlanes); F(c->dMapRGB); F(c->dLatticeOwner); F(c->dTile4); F(c->dLoaded);
        c->dw = w; c->dh = h;
        const int tileW = w >> 3, tileH = h >> 3;
        c->dPlaneSize = (size_t)tileW * tileH * 64;
        const int stride4 = (w + 15) >> 4;
        c->dTile4Size = (size_t)((stride4 << 2) * (((h + 7) >> 3) << 1)) >> 3;
        YK_HIP(c, hipMalloc(&c->dPlanes, c->dPlaneSize * 3));
        YK_HIP(c, hipMalloc(&c->dMapRGB, lat * 3 + 4));                              // + 4: the render reads a corner's three bytes as one word
        YK_HIP(c, hipMalloc(&c->dLatticeOwner, lat * 4));
        YK_HIP(c, hipMalloc(&c->dLoaded, lat));
        YK_HIP(c, hipMalloc(&c->dTile4, ((3 * c->dTile4Size + 3) & ~(size_t)3) + 4));   // three planes once the masks are split
    }
    // The planes (3 B per pixel: 201 MB at 8192 x 8192, 40 us of every frame) are NOT cleared: every renderer marks what it writes in tile4x4Mask and
    // the 1-D decode writes every unmarked quadrant, so after a whole file nothing stale is left; a flow that stops earlier (or writes without
    // marking: the plane-subset loops) gets the unmarked cells zeroed when it needs them (yk_dec_settle).
    c->dPlanesStale = true;
    YK_HIP(c, hipMemsetAsync(c->dMapRGB, 0, lat * 3, c->stream));
    YK_HIP(c, hipMemsetAsync(c->dLoaded, 0, lat, c->stream));
    YK_HIP(c, hipMemsetAsync(c->dTile4, 0, ((3 * c->dTile4Size + 3) & ~(size_t)3) + 4, c->stream));
    c->dSplit = false;
    return YK_OK;
}

// PaletteFullRangeRemapping (decoder/YAIK_GenericFunctions.cpp:128-137) of a corner stream that is still in HBM: v * ((255 << 16) / range) >> 16
__global__ void yk_dec_remap_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t n, uint32_t factor) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (uint8_t)(((uint32_t)src[i] * factor) >> 16);
}

static int yk_decode_gradient_impl(yk_ctx* c, int sx, int sy, const uint8_t* bitmap, size_t bitmapBytes, const uint8_t* rgb, size_t rgbBytes, bool onDevice, int remapRange) {
    if (!c || !bitmap) return YK_ERR_BAD_ARG;
    if (!c->dPlanes) return yk_fail(c, YK_ERR_STATE, "yk_decode_begin first");
    static const int ok[7][2] = { {4,4},{4,3},{3,4},{3,3},{3,2},{2,3},{2,2} };
    bool found = false; for (auto& o : ok) found |= (o[0] == sx && o[1] == sy);
    if (!found) return yk_fail(c, YK_ERR_BAD_ARG, "unsupported tile format");
    YK_HIP(c, hipSetDevice(c->device));
    const int w = c->dw, h = c->dh, latW = w / 4 + 1;
    const size_t lat = (size_t)latW * (h / 4 + 1);
    const DPassGeo g = yk_dpass_geo(sx, sy, w);
    const size_t need = ((size_t)g.xBB * ((h + g.bigY - 1) / g.bigY) * g.bitCount) >> 3;
    if (bitmapBytes < need) return yk_fail(c, YK_ERR_RANGE, "tile bitmap shorter than the image needs");
    const size_t nWords = (need + 3) / 4, nBits = need * 8, nb = (nBits + 1023) / 1024;
    // scratch: [bitmap words][rgb][blockSums][total]
    const size_t oB = 0, oR = oB + nWords * 4 + 16, oBB = (oR + rgbBytes + 19) & ~(size_t)15, oT = oBB + nb * 4 + 16;
    int rc = yk_dec_scratch(c, oT + 64); if (rc) return rc;
    uint8_t* S = c->dScratch;
    YK_HIP(c, hipMemsetAsync(S + oB, 0, nWords * 4, c->stream));
    YK_HIP(c, hipMemcpyAsync(S + oB, bitmap, need, onDevice ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    if (rgbBytes) {
        if (onDevice && remapRange > 0)
            hipLaunchKernelGGL(yk_dec_remap_kernel, dim3((unsigned)((rgbBytes + 255) / 256 < 1024 ? (rgbBytes + 255) / 256 : 1024)), dim3(256), 0, c->stream,
                               rgb, S + oR, rgbBytes, (uint32_t)((255u << 16) / (uint32_t)remapRange));
        else YK_HIP(c, hipMemcpyAsync(S + oR, rgb, rgbBytes, onDevice ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    }
    YK_HIP(c, hipMemsetAsync(c->dLatticeOwner, 0xFF, lat * 4, c->stream));
    const uint32_t* bm = reinterpret_cast<const uint32_t*>(S + oB);
    uint32_t* blockSums = reinterpret_cast<uint32_t*>(S + oBB);
    uint32_t* total = reinterpret_cast<uint32_t*>(S + oT);
    { int rc2 = yk_stage_begin(c, YK_STAGE_DEC_GRADIENT); if (rc2) return rc2; }
    hipLaunchKernelGGL(yk_dec_owner_kernel, dim3((unsigned)((nBits + 255) / 256)), dim3(256), 0, c->stream, bm, nBits, g, w, h, latW, c->dLoaded, c->dLatticeOwner);
    hipLaunchKernelGGL(yk_dec_corner_kernel<false>, dim3((unsigned)nb), dim3(1024), 0, c->stream, bm, nBits, g, w, h, latW, c->dLoaded, c->dLatticeOwner,
                       blockSums, (const uint8_t*)nullptr, (size_t)0, (uint8_t*)nullptr);
    hipLaunchKernelGGL(yk_u32_scanblocks_kernel, dim3(1), dim3(1024), 0, c->stream, blockSums, (int)nb, total);
    hipLaunchKernelGGL(yk_dec_corner_kernel<true>, dim3((unsigned)nb), dim3(1024), 0, c->stream, bm, nBits, g, w, h, latW, c->dLoaded, c->dLatticeOwner,
                       blockSums, S + oR, rgbBytes, c->dMapRGB);
    hipLaunchKernelGGL(yk_dec_render_kernel, dim3((unsigned)nWords), dim3(256), 0, c->stream, bm, nWords, g, w, h, latW, c->dMapRGB, c->dPlanes, c->dPlaneSize,
                       w >> 3, reinterpret_cast<uint32_t*>(c->dTile4), (w + 15) >> 4);
    YK_HIP(c, hipGetLastError());
    { int rc2 = yk_stage_end(c, YK_STAGE_DEC_GRADIENT); if (rc2) return rc2; }
    if (!onDevice) YK_HIP(c, hipStreamSynchronize(c->stream));            // the host buffers may be reused by the caller
    return YK_OK;
}

int yk_decode_gradient(yk_ctx* c, int sx, int sy, const uint8_t* bitmap, size_t bitmapBytes, const uint8_t* rgb, size_t rgbBytes) {
    return yk_decode_gradient_impl(c, sx, sy, bitmap, bitmapBytes, rgb, rgbBytes, false, 0);
}

int yk_decode_gradient_device(yk_ctx* c, int sx, int sy, const uint8_t* devBitmap, size_t bitmapBytes, const uint8_t* devRgb, size_t rgbBytes, int remapRange) {
    return yk_decode_gradient_impl(c, sx, sy, devBitmap, bitmapBytes, devRgb, rgbBytes, true, remapRange);
}

int yk_decode_gradient_all_device(yk_ctx* c, int nPasses, const int* tileShiftX, const int* tileShiftY, const uint8_t* const* devBitmap, const size_t* bitmapBytes,
                                  const uint8_t* const* devRgb, const size_t* rgbBytes, int remapRange) {
    if (!c || nPasses < 0 || (nPasses && (!tileShiftX || !tileShiftY || !devBitmap || !bitmapBytes || !devRgb || !rgbBytes))) return YK_ERR_BAD_ARG;
    if (!c->dPlanes) return yk_fail(c, YK_ERR_STATE, "yk_decode_begin first");
    if (nPasses == 0) return YK_OK;
    const int w = c->dw, h = c->dh, latW = w / 4 + 1;
    const size_t lat = (size_t)latW * (h / 4 + 1);
    static const int ok[7][2] = { {4,4},{4,3},{3,4},{3,3},{3,2},{2,3},{2,2} };
    bool oneCall = nPasses <= 7;
    size_t need[7] = {};
    for (int p = 0; p < nPasses; p++) {
        bool found = false; for (auto& o : ok) found |= (o[0] == tileShiftX[p] && o[1] == tileShiftY[p]);
        if (!found) return yk_fail(c, YK_ERR_BAD_ARG, "unsupported tile format");
        if (!devBitmap[p] || (rgbBytes[p] && !devRgb[p])) return YK_ERR_BAD_ARG;
        if (p < 7) {
            const DPassGeo g = yk_dpass_geo(tileShiftX[p], tileShiftY[p], w);
            need[p] = ((size_t)g.xBB * ((h + g.bigY - 1) / g.bigY) * g.bitCount) >> 3;
            if (bitmapBytes[p] < need[p]) return yk_fail(c, YK_ERR_RANGE, "tile bitmap shorter than the image needs");
            if (need[p] * 8 >= ((size_t)1 << 25) || rgbBytes[p] > 0xFFFFFFFFu) oneCall = false;      // the key holds 25 bits of tile position
        }
    }
    if (!oneCall) {                                                          // very large images / more than seven chunks: pass after pass
        for (int p = 0; p < nPasses; p++) {
            const int rc = yk_decode_gradient_impl(c, tileShiftX[p], tileShiftY[p], devBitmap[p], bitmapBytes[p], devRgb[p], rgbBytes[p], true, remapRange);
            if (rc) return rc;
        }
        return YK_OK;
    }
    YK_HIP(c, hipSetDevice(c->device));
    DecPlan pl = {};
    pl.n = nPasses;
    for (int p = 0; p < 7; p++) {
        const size_t nb = p < nPasses ? need[p] : 0;
        pl.nBytes[p] = (uint32_t)nb; pl.rgbBytes[p] = p < nPasses ? (uint32_t)rgbBytes[p] : 0u;
        pl.bm[p] = p < nPasses ? devBitmap[p] : nullptr; pl.rgb[p] = p < nPasses ? devRgb[p] : nullptr;
        pl.sx[p] = p < nPasses ? tileShiftX[p] : 4; pl.sy[p] = p < nPasses ? tileShiftY[p] : 4;
        pl.byteStart[p + 1] = pl.byteStart[p] + (uint32_t)nb;
        pl.blockStart[p + 1] = pl.blockStart[p] + (uint32_t)((nb + 1023) / 1024);
        pl.wordStart[p + 1] = pl.wordStart[p] + (uint32_t)((nb + 3) / 4);
    }
    const size_t nbTot = pl.blockStart[7], nBytesTot = pl.byteStart[7], nWordsTot = pl.wordStart[7];
    // scratch: [corners per block | words per block | words listed per pass | ownership bits per bitmap byte (one word) | render lists]
    const size_t oS = 0, oW = oS + nbTot * 4, oP = oW + nbTot * 4, oT = oP + 64, oL = (oT + nBytesTot * 4 + 15) & ~(size_t)15;
    { const int rc = yk_dec_scratch(c, oL + nWordsTot * 4 + 64); if (rc) return rc; }
    uint8_t* S = c->dScratch;
    uint32_t* blockSums = reinterpret_cast<uint32_t*>(S + oS);
    uint32_t* blockWords = reinterpret_cast<uint32_t*>(S + oW);
    uint32_t* passWords = reinterpret_cast<uint32_t*>(S + oP);
    uint32_t* perThread = reinterpret_cast<uint32_t*>(S + oT);
    uint32_t* wordList = reinterpret_cast<uint32_t*>(S + oL);
    const uint32_t factor = remapRange > 0 ? (uint32_t)((255u << 16) / (uint32_t)remapRange) : 0u;
    YK_HIP(c, hipMemsetAsync(c->dLatticeOwner, 0xFF, lat * 4, c->stream));
    { int rc2 = yk_stage_begin(c, YK_STAGE_DEC_GRADIENT); if (rc2) return rc2; }
    // 1 clear + 4 launches for all passes + one render per pass (the per-pass form: 3 copies / clears + 5 launches per pass)
    hipLaunchKernelGGL(yk_decall_owner_kernel, dim3((unsigned)((nBytesTot + 255) / 256)), dim3(256), 0, c->stream, pl, w, h, latW, c->dLoaded, c->dLatticeOwner);
    hipLaunchKernelGGL(yk_decall_stream_kernel<false>, dim3((unsigned)nbTot), dim3(1024), 0, c->stream, pl, w, h, latW, c->dLoaded, c->dLatticeOwner, blockSums, blockWords,
                       perThread, (uint8_t*)nullptr, (uint32_t*)nullptr, 0u);
    hipLaunchKernelGGL(yk_decall_scan_kernel, dim3(1), dim3(1024), 0, c->stream, blockSums, blockWords, pl, passWords);
    hipLaunchKernelGGL(yk_decall_stream_kernel<true>, dim3((unsigned)nbTot), dim3(1024), 0, c->stream, pl, w, h, latW, c->dLoaded, c->dLatticeOwner, blockSums, blockWords,
                       perThread, c->dMapRGB, wordList, factor);
    // one render launch: a workgroup per 64x64 block of the image walks the passes in call order
    hipLaunchKernelGGL(yk_decall_render_blocks_kernel, dim3((unsigned)(((w + 63) / 64) * ((h + 63) / 64))), dim3(256), 0, c->stream, pl, w, h, latW,
                       c->dMapRGB, c->dPlanes, c->dPlaneSize, w >> 3, reinterpret_cast<uint32_t*>(c->dTile4), (w + 15) >> 4);
    YK_HIP(c, hipGetLastError());
    { int rc2 = yk_stage_end(c, YK_STAGE_DEC_GRADIENT); if (rc2) return rc2; }
    return YK_OK;
}

static int yk_dec_split(yk_ctx* c) {                                       // UpdateTileAndRGBMask (YAIK_API.cpp:530-544), once
    if (c->dSplit) return YK_OK;
    const size_t lat = (size_t)(c->dw / 4 + 1) * (c->dh / 4 + 1), n = lat > c->dTile4Size ? lat : c->dTile4Size;
    hipLaunchKernelGGL(yk_dec_split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->dLoaded, lat, c->dTile4, c->dTile4Size);
    YK_HIP(c, hipGetLastError());
    c->dSplit = true;
    return YK_OK;
}

int yk_decode_split_masks(yk_ctx* c) {
    if (!c) return YK_ERR_BAD_ARG;
    if (!c->dPlanes) return yk_fail(c, YK_ERR_STATE, "yk_decode_begin first");
    YK_HIP(c, hipSetDevice(c->device));
    return yk_dec_split(c);
}

int yk_decode_gradient_planes(yk_ctx* c, int planeBit, int consistentMarks, const uint8_t* bitmap, size_t bitmapBytes, const uint8_t* rgb, size_t rgbBytes) {
    if (!c || !bitmap) return YK_ERR_BAD_ARG;
    if (planeBit == 7) return yk_decode_gradient(c, 2, 2, bitmap, bitmapBytes, rgb, rgbBytes);
    if (planeBit < 1 || planeBit > 6) return yk_fail(c, YK_ERR_BAD_ARG, "planeBit must be 1..7");
    if (!c->dPlanes) return yk_fail(c, YK_ERR_STATE, "yk_decode_begin first");
    YK_HIP(c, hipSetDevice(c->device));
    { const int rcs = yk_dec_settle(c); if (rcs) return rcs; }               // the plane-subset loops write without marking (reference behaviour): from here on the planes are fully defined
    const int w = c->dw, h = c->dh, latW = w / 4 + 1;
    const size_t lat = (size_t)latW * (h / 4 + 1);
    const size_t need = ((size_t)((w + 31) / 32) * ((h + 31) / 32) * 64) >> 3;
    if (bitmapBytes < need) return yk_fail(c, YK_ERR_RANGE, "tile bitmap shorter than the image needs");
    const size_t nWords = (need + 3) / 4, nBits = need * 8, nb = (nBits + 1023) / 1024;
    const size_t oB = 0, oR = oB + nWords * 4 + 16, oBB = (oR + rgbBytes + 19) & ~(size_t)15, oT = oBB + nb * 4 + 16;
    int rc = yk_dec_scratch(c, oT + 64); if (rc) return rc;
    rc = yk_dec_split(c); if (rc) return rc;                                 // a chunk whose plane field is not 7 splits the masks first (:875-877)
    uint8_t* S = c->dScratch;
    YK_HIP(c, hipMemsetAsync(S + oB, 0, nWords * 4, c->stream));
    YK_HIP(c, hipMemcpyAsync(S + oB, bitmap, need, hipMemcpyHostToDevice, c->stream));
    if (rgbBytes) YK_HIP(c, hipMemcpyAsync(S + oR, rgb, rgbBytes, hipMemcpyHostToDevice, c->stream));
    YK_HIP(c, hipMemsetAsync(c->dLatticeOwner, 0xFF, lat * 4, c->stream));
    const uint32_t* bm = reinterpret_cast<const uint32_t*>(S + oB);
    uint32_t* blockSums = reinterpret_cast<uint32_t*>(S + oBB);
    uint32_t* total = reinterpret_cast<uint32_t*>(S + oT);
    const unsigned g256 = (unsigned)((nBits + 255) / 256);
    hipLaunchKernelGGL(yk_dec44p_owner_kernel, dim3(g256), dim3(256), 0, c->stream, bm, nBits, w, h, latW, planeBit, c->dLoaded, c->dLatticeOwner);
    hipLaunchKernelGGL(yk_dec44p_corner_kernel<false>, dim3((unsigned)nb), dim3(1024), 0, c->stream, bm, nBits, w, h, latW, planeBit, c->dLoaded, c->dLatticeOwner,
                       blockSums, (const uint8_t*)nullptr, (size_t)0, (uint8_t*)nullptr);
    hipLaunchKernelGGL(yk_u32_scanblocks_kernel, dim3(1), dim3(1024), 0, c->stream, blockSums, (int)nb, total);
    hipLaunchKernelGGL(yk_dec44p_corner_kernel<true>, dim3((unsigned)nb), dim3(1024), 0, c->stream, bm, nBits, w, h, latW, planeBit, c->dLoaded, c->dLatticeOwner,
                       blockSums, S + oR, rgbBytes, c->dMapRGB);
    hipLaunchKernelGGL(yk_dec44p_mark_kernel, dim3((unsigned)((lat + 255) / 256)), dim3(256), 0, c->stream, c->dLatticeOwner, lat, planeBit, c->dLoaded);
    hipLaunchKernelGGL(yk_dec44p_render_kernel, dim3(g256), dim3(256), 0, c->stream, bm, nBits, w, h, latW, planeBit, consistentMarks, c->dMapRGB, c->dPlanes,
                       c->dPlaneSize, w >> 3, reinterpret_cast<uint32_t*>(c->dTile4), c->dTile4Size, (w + 15) >> 4);
    YK_HIP(c, hipGetLastError());
    YK_HIP(c, hipStreamSynchronize(c->stream));
    return YK_OK;
}

__global__ void yk_dec1d_next_plane_kernel(uint32_t* __restrict__ runBase, const uint32_t* __restrict__ tot) { if (threadIdx.x < 2) runBase[threadIdx.x] += tot[threadIdx.x]; }

static int yk_decode_1d_impl(yk_ctx* c, const uint8_t* typeStream, size_t typeBytes, const uint8_t* pixStream, size_t pixBytes, int compressionRange, bool onDevice) {
    if (!c || !typeStream || !pixStream || compressionRange <= 0) return YK_ERR_BAD_ARG;
    if (!c->dPlanes) return yk_fail(c, YK_ERR_STATE, "yk_decode_begin first");
    YK_HIP(c, hipSetDevice(c->device));
    const int w = c->dw, h = c->dh, tilesW = w >> 3;
    const size_t T8 = (size_t)tilesW * (h >> 3), nb = (T8 + 1023) / 1024;
    const size_t oTy = 0, oPx = (oTy + typeBytes + 31) & ~(size_t)15, oCT = (oPx + pixBytes + 31) & ~(size_t)15, oCP = oCT + 16,
                 oBT = oCP + 16, oBP = oBT + nb * 4 + 16, oTot = oBP + nb * 4 + 16, oOff = oTot + 64;
    int rc = yk_dec_scratch(c, oOff + T8 * 4 + 64); if (rc) return rc;
    uint8_t* S = c->dScratch;
    // streams that already lie in HBM (16-byte aligned, as the encoder leaves them) are read where they are
    const bool inPlace = onDevice && ((reinterpret_cast<uintptr_t>(typeStream) | reinterpret_cast<uintptr_t>(pixStream)) & 15) == 0;
    if (!inPlace) {
        YK_HIP(c, hipMemcpyAsync(S + oTy, typeStream, typeBytes, onDevice ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
        YK_HIP(c, hipMemcpyAsync(S + oPx, pixStream, pixBytes, onDevice ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    }
    const uint8_t* const tyS = inPlace ? typeStream : S + oTy; const uint8_t* const pxS = inPlace ? pixStream : S + oPx;
    uint32_t* bT = reinterpret_cast<uint32_t*>(S + oBT); uint32_t* bP = reinterpret_cast<uint32_t*>(S + oBP);
    uint32_t* tot = reinterpret_cast<uint32_t*>(S + oTot);
    uint32_t* offInBlk = reinterpret_cast<uint32_t*>(S + oOff);
    const unsigned gTiles = (unsigned)((T8 + 64 * YK_D1_TPL - 1) / (64 * YK_D1_TPL));
    { int rc2 = yk_stage_begin(c, YK_STAGE_DEC_1D); if (rc2) return rc2; }
    if (!c->dSplit) {
        // no partial-plane pass ran: the three planes share one mask, one count / scan serves all of them (3 launches)
        hipLaunchKernelGGL(yk_dec1d_count_kernel, dim3((unsigned)nb), dim3(1024), 0, c->stream, c->dTile4, (w + 15) >> 4, tilesW, T8, bT, bP, offInBlk);
        hipLaunchKernelGGL(yk_dec1d_scan_kernel, dim3(1), dim3(1024), 0, c->stream, bT, bP, (int)nb, tot);
        hipLaunchKernelGGL(yk_dec1d_kernel, dim3(gTiles, 3), dim3(256), 0, c->stream, (const uint32_t*)offInBlk, T8, bT, bP, tot,
                           tyS, typeBytes, pxS, pixBytes, (1 << 24) / compressionRange, c->dPlanes, c->dPlaneSize, -1, (const uint32_t*)nullptr);
    } else {
        // per-plane masks (Decompress1D reads tile4x4Mask + planeID * tile4x4MaskSize, YAIK_3DTile.cpp:41): one plane after the other,
        // every plane's streams start where the plane before it stopped
        uint32_t* runBase = tot + 4;
        YK_HIP(c, hipMemsetAsync(runBase, 0, 2 * sizeof(uint32_t), c->stream));
        for (int p = 0; p < 3; p++) {
            const uint8_t* t4 = c->dTile4 + (size_t)p * c->dTile4Size;
            hipLaunchKernelGGL(yk_dec1d_count_kernel, dim3((unsigned)nb), dim3(1024), 0, c->stream, t4, (w + 15) >> 4, tilesW, T8, bT, bP, offInBlk);
            hipLaunchKernelGGL(yk_dec1d_scan_kernel, dim3(1), dim3(1024), 0, c->stream, bT, bP, (int)nb, tot);
            hipLaunchKernelGGL(yk_dec1d_kernel, dim3(gTiles, 1), dim3(256), 0, c->stream, (const uint32_t*)offInBlk, T8, bT, bP, tot,
                               tyS, typeBytes, pxS, pixBytes, (1 << 24) / compressionRange, c->dPlanes, c->dPlaneSize, p, (const uint32_t*)runBase);
            hipLaunchKernelGGL(yk_dec1d_next_plane_kernel, dim3(1), dim3(64), 0, c->stream, runBase, (const uint32_t*)tot);
        }
    }
    YK_HIP(c, hipGetLastError());
    { int rc2 = yk_stage_end(c, YK_STAGE_DEC_1D); if (rc2) return rc2; }
    c->dPlanesStale = false;                                                 // every unmarked quadrant of every plane has been written
    if (!onDevice) YK_HIP(c, hipStreamSynchronize(c->stream));
    return YK_OK;
}

int yk_decode_1d(yk_ctx* c, const uint8_t* typeStream, size_t typeBytes, const uint8_t* pixStream, size_t pixBytes, int compressionRange) {
    return yk_decode_1d_impl(c, typeStream, typeBytes, pixStream, pixBytes, compressionRange, false);
}

int yk_decode_1d_device(yk_ctx* c, const uint8_t* devType, size_t typeBytes, const uint8_t* devPix, size_t pixBytes, int compressionRange) {
    return yk_decode_1d_impl(c, devType, typeBytes, devPix, pixBytes, compressionRange, true);
}

int yk_decode_mask(yk_ctx* c, const uint8_t* bits, int bw, int bh, uint8_t* hostOut, size_t cap) {
    if (!c || !bits || !hostOut || bw <= 0 || bh <= 0) return YK_ERR_BAD_ARG;
    const size_t outBytes = (size_t)bw * bh * 32, inBytes = ((size_t)bw * bh + 7) / 8;
    if (cap < outBytes) return yk_fail(c, YK_ERR_RANGE, "mask buffer too small");
    YK_HIP(c, hipSetDevice(c->device));
    const size_t oO = (inBytes + 31) & ~(size_t)15;
    int rc = yk_dec_scratch(c, oO + outBytes + 64); if (rc) return rc;
    YK_HIP(c, hipMemcpyAsync(c->dScratch, bits, inBytes, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(yk_dec_mask_kernel, dim3((unsigned)((bw * bh + 255) / 256)), dim3(256), 0, c->stream, c->dScratch, bw, bh,
                       reinterpret_cast<unsigned long long*>(c->dScratch + oO));
    YK_HIP(c, hipMemcpyAsync(hostOut, c->dScratch + oO, outBytes, hipMemcpyDeviceToHost, c->stream));
    YK_HIP(c, hipStreamSynchronize(c->stream));
    return YK_OK;
}

int yk_decode_planes(yk_ctx* c, uint8_t* hostR, uint8_t* hostG, uint8_t* hostB, size_t capEach) {
    if (!c || !hostR || !hostG || !hostB) return YK_ERR_BAD_ARG;
    if (!c->dPlanes) return yk_fail(c, YK_ERR_STATE, "yk_decode_begin first");
    if (capEach < c->dPlaneSize) return yk_fail(c, YK_ERR_RANGE, "plane buffer too small");
    YK_HIP(c, hipSetDevice(c->device));
    { const int rcs = yk_dec_settle(c); if (rcs) return rcs; }
    uint8_t* dst[3] = { hostR, hostG, hostB };
    for (int p = 0; p < 3; p++) YK_HIP(c, hipMemcpyAsync(dst[p], c->dPlanes + p * c->dPlaneSize, c->dPlaneSize, hipMemcpyDeviceToHost, c->stream));
    YK_HIP(c, hipStreamSynchronize(c->stream));
    return YK_OK;
}

// Shared body of the two output entry points.  The device image is packed tight (w * bpp bytes per row) and lands in the caller's
// buffer by a 2-D copy with dpitch = outputImageStride: like the reference's loop, bytes of a row beyond the pixels are never
// touched (outputImageStride exists to place the image INSIDE a larger buffer, include/YAIK.h:190).
static int yk_decode_output_impl(yk_ctx* c, uint8_t* hostOut, size_t outputImageStride, const uint8_t* hostAlpha, int strideA, bool refRGBA) {
    if (!c || !hostOut) return YK_ERR_BAD_ARG;
    if (!c->dPlanes) return yk_fail(c, YK_ERR_STATE, "yk_decode_begin first");
    const int w = c->dw, h = c->dh, bpp = (hostAlpha && !refRGBA) ? 4 : 3;
    const size_t rowBytes = (size_t)w * bpp + (refRGBA && hostAlpha ? 1 : 0);
    if (outputImageStride < rowBytes || (hostAlpha && strideA < w)) return yk_fail(c, YK_ERR_BAD_ARG, "output stride too small");
    YK_HIP(c, hipSetDevice(c->device));
    const size_t dPitch = (rowBytes + 15) & ~(size_t)15;
    const size_t outBytes = dPitch * h, aBytes = hostAlpha ? (size_t)strideA * h : 0, oA = (outBytes + 31) & ~(size_t)15;
    int rc = yk_dec_scratch(c, oA + aBytes + 64); if (rc) return rc;
    rc = yk_dec_settle(c); if (rc) return rc;
    if (hostAlpha) YK_HIP(c, hipMemcpyAsync(c->dScratch + oA, hostAlpha, aBytes, hipMemcpyHostToDevice, c->stream));
    { int rc2 = yk_stage_begin(c, YK_STAGE_DEC_DETILE); if (rc2) return rc2; }
    hipLaunchKernelGGL(yk_dec_detile_kernel, dim3((w + 255) / 256, (h + 3) / 4), dim3(256), 0, c->stream, c->dPlanes, c->dPlaneSize, w >> 3, w, h,
                       (hostAlpha && !refRGBA) ? c->dScratch + oA : (const uint8_t*)nullptr, strideA, c->dScratch, dPitch);
    YK_HIP(c, hipGetLastError());
    { int rc2 = yk_stage_end(c, YK_STAGE_DEC_DETILE); if (rc2) return rc2; }
    if (refRGBA && hostAlpha) {
        hipLaunchKernelGGL(yk_dec_ref_alpha_kernel, dim3((h + 255) / 256), dim3(256), 0, c->stream, c->dScratch + oA, strideA, aBytes, w, h, c->dScratch, dPitch);
        YK_HIP(c, hipGetLastError());
    }
    YK_HIP(c, hipMemcpy2DAsync(hostOut, outputImageStride, c->dScratch, dPitch, rowBytes, (size_t)h, hipMemcpyDeviceToHost, c->stream));
    YK_HIP(c, hipStreamSynchronize(c->stream));
    return YK_OK;
}

int yk_decode_output(yk_ctx* c, uint8_t* hostOut, size_t outputImageStride, const uint8_t* hostAlpha, int strideA) {
    return yk_decode_output_impl(c, hostOut, outputImageStride, hostAlpha, strideA, false);
}

int yk_decode_output_reference_rgba(yk_ctx* c, uint8_t* hostOut, size_t outputImageStride, const uint8_t* hostAlpha, int strideA) {
    return yk_decode_output_impl(c, hostOut, outputImageStride, hostAlpha, strideA, true);
}

const uint8_t* yk_decode_planes_device(yk_ctx* c, size_t* planeSize) {
    if (!c || !c->dPlanes) return nullptr;
    if (hipSetDevice(c->device) != hipSuccess || yk_dec_settle(c) != YK_OK) return nullptr;
    if (planeSize) *planeSize = c->dPlaneSize;
    return c->dPlanes;
}

int yk_decode_tile4x4_planes(yk_ctx* c, uint8_t* hostOut, size_t cap) {     // the three planes of tile4x4Mask (planes 1, 2 are meaningful once split)
    if (!c || !hostOut) return YK_ERR_BAD_ARG;
    if (!c->dPlanes) return yk_fail(c, YK_ERR_STATE, "yk_decode_begin first");
    if (cap < 3 * c->dTile4Size) return yk_fail(c, YK_ERR_RANGE, "tile4x4 buffer too small");
    YK_HIP(c, hipSetDevice(c->device));
    YK_HIP(c, hipMemcpyAsync(hostOut, c->dTile4, 3 * c->dTile4Size, hipMemcpyDeviceToHost, c->stream));
    YK_HIP(c, hipStreamSynchronize(c->stream));
    return YK_OK;
}

int yk_decode_tile4x4(yk_ctx* c, uint8_t* hostOut, size_t cap) {
    if (!c || !hostOut) return YK_ERR_BAD_ARG;
    if (!c->dPlanes) return yk_fail(c, YK_ERR_STATE, "yk_decode_begin first");
    if (cap < c->dTile4Size) return yk_fail(c, YK_ERR_RANGE, "tile4x4 buffer too small");
    YK_HIP(c, hipSetDevice(c->device));
    YK_HIP(c, hipMemcpyAsync(hostOut, c->dTile4, c->dTile4Size, hipMemcpyDeviceToHost, c->stream));
    YK_HIP(c, hipStreamSynchronize(c->stream));
    return YK_OK;
}

}  // extern "C"
