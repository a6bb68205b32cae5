// micro-benchmark: do fp32 VALU ops issue faster than int32 VALU ops on gfx950?  8 waves/SIMD, 8 independent chains per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void k(float* out, float seed, int iters) {
    float a[8]; unsigned u[8];
    for (int i = 0; i < 8; i++) { a[i] = seed * (threadIdx.x + 1) + i * 0.37f; u[i] = (unsigned)(threadIdx.x * 97 + i); }
    float b = seed * 0.5f + 1.0f; unsigned ub = 0x1234567u ^ (unsigned)threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0) a[i] = a[i] - b;                                                        // v_sub_f32
                else if (KIND == 1) a[i] = __builtin_fmaf(a[i], b, (float)r);                            // v_fma_f32
                else if (KIND == 2) { float t; asm volatile("v_min3_f32 %0, %1, |%2|, |%3|" : "=v"(t) : "v"(a[i]), "v"(b), "v"(a[(i + 1) & 7])); a[i] = t; }
                else if (KIND == 3) { float t; asm volatile("v_min_f32 %0, %1, |%2|" : "=v"(t) : "v"(a[i]), "v"(b)); a[i] = t; }
                else if (KIND == 4) u[i] = u[i] + ub;                                                   // v_add_u32
                else if (KIND == 5) { unsigned t; asm volatile("v_min_u32 %0, %1, %2" : "=v"(t) : "v"(u[i]), "v"(ub)); u[i] = t; }
                else if (KIND == 6) { unsigned t; asm volatile("v_cvt_pk_u8_f32 %0, %1, %2, %3" : "=v"(t) : "v"(a[i]), "v"(ub), "v"(u[i])); u[i] = t; }
                else if (KIND == 7) { float t; asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(*(double*)&a[i & 6]) : "v"(*(double*)&a[i & 6]), "v"(*(double*)&a[(i + 2) & 6])); (void)t; }
                else if (KIND == 8) { unsigned t; asm volatile("v_sad_u32 %0, %1, %2, %3" : "=v"(t) : "v"(u[i]), "v"(ub), "v"(u[(i+1)&7])); u[i] = t; }
            }
    }
    float s = 0; for (int i = 0; i < 8; i++) s += a[i] + (float)u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, (1 << 22) * 4);
    const char* names[9] = { "v_sub_f32", "v_fma_f32", "v_min3_f32 |abs|", "v_min_f32 |abs|", "v_add_u32", "v_min_u32", "v_cvt_pk_u8_f32", "v_pk_add_f32 (2 lanes-ops)", "v_sad_u32" };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int kind = 0; kind < 9; kind++) {
        int iters = 4000;
        dim3 g(256 * 8), b(256);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            switch (kind) {
            case 0: hipLaunchKernelGGL(k<0>, g, b, 0, 0, d, 1.5f, iters); break; case 1: hipLaunchKernelGGL(k<1>, g, b, 0, 0, d, 1.5f, iters); break;
            case 2: hipLaunchKernelGGL(k<2>, g, b, 0, 0, d, 1.5f, iters); break; case 3: hipLaunchKernelGGL(k<3>, g, b, 0, 0, d, 1.5f, iters); break;
            case 4: hipLaunchKernelGGL(k<4>, g, b, 0, 0, d, 1.5f, iters); break; case 5: hipLaunchKernelGGL(k<5>, g, b, 0, 0, d, 1.5f, iters); break;
            case 6: hipLaunchKernelGGL(k<6>, g, b, 0, 0, d, 1.5f, iters); break; case 7: hipLaunchKernelGGL(k<7>, g, b, 0, 0, d, 1.5f, iters); break;
            case 8: hipLaunchKernelGGL(k<8>, g, b, 0, 0, d, 1.5f, iters); break; }
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double waveInstr = 8192.0 * iters * 128.0;
        printf("%-28s %.3f ms -> %.3f T wave-instr/s\n", names[kind], ms, waveInstr / (ms * 1e-3) / 1e12);
    }
    return 0;
}
