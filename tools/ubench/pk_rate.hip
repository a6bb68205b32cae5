// micro-benchmark 3: packed 16-bit / packed fp32 VALU issue rates on gfx950 (wave-instructions per second, whole chip)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void k(uint32_t* out, uint32_t seed, int iters) {
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = seed * (threadIdx.x + 1) + i * 0x00010001u;
    uint32_t b = seed * 3 + 0x3c003c00u, c = seed + 0x3c003c00u;
    uint64_t a2[4];
    for (int i = 0; i < 4; i++) a2[i] = ((uint64_t)a[i] << 32) | a[i + 4];
    uint64_t b2 = ((uint64_t)b << 32) | c;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                else if (KIND == 1) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                else if (KIND == 2) asm volatile("v_pk_fma_f16 %0, %0, %1, %2 clamp" : "+v"(a[i]) : "v"(b), "v"(c));
                else if (KIND == 3) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                else if (KIND == 4) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                else if (KIND == 5) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(a[i]) : "v"(b));
                else if (KIND == 6) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                else if (KIND == 7) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a2[i & 3]) : "v"(b2));
                else if (KIND == 8) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a2[i & 3]) : "v"(b2));
                else if (KIND == 9) asm volatile("v_add_f16_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                else if (KIND == 10) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                else if (KIND == 11) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                else if (KIND == 12) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                else if (KIND == 13) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            }
    }
    uint32_t s = 0; for (int i = 0; i < 8; i++) s += a[i];
    for (int i = 0; i < 4; i++) s += (uint32_t)a2[i] + (uint32_t)(a2[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND> static void run(uint32_t* d, const char* name, hipEvent_t e0, hipEvent_t e1) {
    const int iters = 4000;
    dim3 g(256 * 8), b(256);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, g, b, 0, 0, d, 3u, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr = 8192.0 * iters * 128.0;          // wave-instructions
    printf("%-28s %.3f ms -> %.3f T wave-instr/s\n", name, ms, instr / (ms * 1e-3) / 1e12);
}
int main() {
    uint32_t* d; hipMalloc(&d, (1 << 22) * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    run<0>(d, "v_pk_add_f16", e0, e1);
    run<1>(d, "v_pk_fma_f16", e0, e1);
    run<2>(d, "v_pk_fma_f16 clamp", e0, e1);
    run<3>(d, "v_pk_add_u16", e0, e1);
    run<4>(d, "v_pk_min_u16", e0, e1);
    run<5>(d, "v_pk_sub_u16 clamp", e0, e1);
    run<6>(d, "v_pk_mad_u16", e0, e1);
    run<7>(d, "v_pk_fma_f32", e0, e1);
    run<8>(d, "v_pk_add_f32", e0, e1);
    run<9>(d, "v_add_f16", e0, e1);
    run<10>(d, "v_pk_max_f16", e0, e1);
    run<11>(d, "v_dot2_f32_f16", e0, e1);
    run<12>(d, "v_pk_mul_f16", e0, e1);
    run<13>(d, "v_pk_min_i16", e0, e1);
    return 0;
}
