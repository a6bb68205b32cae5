// Device-side helpers shared by the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// exclusive scan of one uint32 per thread over a 1024-thread block (16 wave64); *total = block sum
__device__ __forceinline__ uint32_t yk_block_exscan(uint32_t v, uint32_t* s_tmp, uint32_t* total) {
    // exclusive scan over the 1024 threads of a block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t y = __shfl_up(x, d); if (lane >= d) x += y; }
    if (lane == 63) s_tmp[wave] = x;
    __syncthreads();
    if (wave == 0) {
        uint32_t t = lane < 16 ? s_tmp[lane] : 0;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) { uint32_t y = __shfl_up(t, d); if (lane >= d) t += y; }
        if (lane < 16) s_tmp[16 + lane] = t;
    }
    __syncthreads();
    const uint32_t waveBase = wave ? s_tmp[16 + wave - 1] : 0;
    *total = s_tmp[16 + 15];
    __syncthreads();
    return waveBase + x - v;
}

// generic two-level exclusive scan over a uint32 array (n <= 1024*1024*... elements, block = 1024 items):
//   yk_u32_blocksum_kernel -> yk_u32_scanblocks_kernel -> consumer adds blockBase[blockIdx.x] to its local exclusive scan
__global__ __launch_bounds__(1024) static void yk_u32_blocksum_kernel(const uint32_t* __restrict__ in, size_t n, uint32_t* __restrict__ blockSums) {
    __shared__ uint32_t s_tmp[32];
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    uint32_t tot;
    yk_block_exscan(i < n ? in[i] : 0u, s_tmp, &tot);
    if (threadIdx.x == 0) blockSums[blockIdx.x] = tot;
}

// in place: blockSums[b] <- exclusive prefix; *total <- grand total (added to *total's previous value if accumulate)
__global__ __launch_bounds__(1024) static void yk_u32_scanblocks_kernel(uint32_t* __restrict__ blockSums, int nBlocks, uint32_t* __restrict__ total) {
    __shared__ uint32_t s_tmp[32];
    uint32_t base = 0;
    for (int start = 0; start < nBlocks; start += 1024) {
        const int i = start + threadIdx.x;
        const uint32_t v = i < nBlocks ? blockSums[i] : 0u;
        uint32_t tot;
        const uint32_t e = yk_block_exscan(v, s_tmp, &tot);
        if (i < nBlocks) blockSums[i] = base + e;
        base += tot;
    }
    if (threadIdx.x == 0) *total = base;
}

// n / d for small non-negative integers held in floats, given r = RN(1/d) (or r = 0 -> returns +0): one Markstein
// correction step.  Exhaustively verified against __fdiv_rn for n in 0..255, d in 1..256 by yk_selftest_kernel.
__device__ __forceinline__ float yk_div_exact(float n, float d, float r) {
    const float q0 = __fmul_rn(n, r);
    const float e = __fmaf_rn(-q0, d, n);
    return __fmaf_rn(e, r, q0);
}

// GetValueModel1 (encoder/EncoderContext.cpp:8383-8391) without a division per pixel: a coded pixel's byte is
//     1 + trunc(((v - minCol) * 15 + (delta >> 1) - 1) / delta)  =  floor((n + delta) / delta),  n + delta >= 0 for every delta >= 1,
// and floor(x / delta) == (x * M) >> 20 with M = floor(2^20 / delta) + 1 for every x the path can produce (x <= 16.5 delta <= 4207: the
// error term x * (M * delta - 2^20) stays below 2^20; exhaustive over delta and v - minCol: yk_selftest 4).  Folded into ONE 24-bit
// multiply-add per pixel: byte = (v * A + B) >> 20 with A = 15 M < 2^24 and B = ((delta >> 1) - 1 + delta - 15 minCol) * M mod 2^32.
// delta == 0 (one value left besides color0 +- 1): every coded pixel is 1.
__device__ __forceinline__ void yk_r1_magic(int delta, int minCol, uint32_t* A, uint32_t* B) {
    if (delta == 0) { *A = 0u; *B = 1u << 20; return; }
    int m = __float2int_rz(1048576.0f * __builtin_amdgcn_rcpf((float)delta));     // floor(2^20 / delta) +- 1
    int r = (1 << 20) - m * delta;
    if (r < 0) { m--; r += delta; }
    if (r >= delta) m++;
    const uint32_t M = (uint32_t)m + 1u;
    *A = 15u * M;
    *B = (uint32_t)((delta >> 1) - 1 + delta - 15 * minCol) * M;
}
