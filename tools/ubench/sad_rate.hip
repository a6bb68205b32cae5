// micro-benchmark: relative cost of v_sad_u32 / v_min3_u32 / v_mqsad_pk_u16_u8 / v_pk_min_u16 on gfx950.
// Each kernel runs the same dependent-chain structure (8 independent chains per lane); whole-kernel wall time is compared.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
template <int KIND>
__global__ void k(unsigned* out, unsigned seed, int iters) {
    unsigned a[8]; u64 q[8];
    for (int i = 0; i < 8; i++) { a[i] = seed * (threadIdx.x + 1) + i * 7919u; q[i] = (u64)a[i] * 0x9E3779B97F4A7C15ULL; }
    unsigned b = seed ^ 0x55aa;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0) a[i] = (a[i] ^ b) + r;                                  // 2 simple ops (xor, add) -> may fuse to v_xad
                else if (KIND == 1) a[i] = __usad(a[i], b, r);                         // v_sad_u32
                else if (KIND == 2) { unsigned t; asm volatile("v_min3_u32 %0, %1, %2, %3" : "=v"(t) : "v"(a[i]), "v"(b), "v"(a[(i + 1) & 7])); a[i] = t + 1; }
                else if (KIND == 3) q[i] = __builtin_amdgcn_mqsad_pk_u16_u8(q[i], b, q[i]);   // v_mqsad_pk_u16_u8
                else if (KIND == 4) { unsigned t; asm volatile("v_pk_min_u16 %0, %1, %2" : "=v"(t) : "v"(a[i]), "v"(b)); a[i] = t + r; }
                else if (KIND == 5) a[i] = a[i] * 3u + b;                              // v_mad_u32_u24 / mul_lo
            }
    }
    unsigned s = 0; for (int i = 0; i < 8; i++) s += a[i] + (unsigned)q[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    unsigned* d; hipMalloc(&d, (1 << 22) * 4);
    const char* names[6] = { "xor+add (2 ops)", "v_sad_u32", "v_min3_u32 + add", "v_mqsad_pk_u16_u8", "v_pk_min_u16 + add", "mul+add" };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int kind = 0; kind < 6; kind++) {
        int iters = 4000;
        dim3 g(256 * 8), b(256);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (kind == 0) hipLaunchKernelGGL(k<0>, g, b, 0, 0, d, 123u, iters);
            if (kind == 1) hipLaunchKernelGGL(k<1>, g, b, 0, 0, d, 123u, iters);
            if (kind == 2) hipLaunchKernelGGL(k<2>, g, b, 0, 0, d, 123u, iters);
            if (kind == 3) hipLaunchKernelGGL(k<3>, g, b, 0, 0, d, 123u, iters);
            if (kind == 4) hipLaunchKernelGGL(k<4>, g, b, 0, 0, d, 123u, iters);
            if (kind == 5) hipLaunchKernelGGL(k<5>, g, b, 0, 0, d, 123u, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // 2048 blocks x 4 waves = 8192 waves = 8 per SIMD; ops per wave = iters*128
        double waveInstr = 8192.0 * iters * 128.0;
        printf("%-22s %.3f ms  -> %.2f ns per 1e6 wave-ops, %.3f T wave-ops/s (statement groups)\n", names[kind], ms, ms * 1e6 / (waveInstr / 1e6), waveInstr / (ms * 1e-3) / 1e12);
    }
    return 0;
}
