// micro-benchmark 2: v_sub_f32 with clamp (VOP3), v_fmac_f32, v_fract_f32, chain of sub_clamp+fma (the threshold-search step)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void k(float* out, float seed, int iters) {
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = seed * (threadIdx.x + 1) + i * 0.37f;
    float b = seed * 0.5f + 1.0f, c = seed + 3.0f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0) { float t; asm volatile("v_sub_f32_e64 %0, %1, %2 clamp" : "=v"(t) : "v"(a[i]), "v"(b)); a[i] = t; }
                else if (KIND == 1) { asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); }
                else if (KIND == 2) { float t; asm volatile("v_fract_f32_e32 %0, %1" : "=v"(t) : "v"(a[i])); a[i] = t; }
                else if (KIND == 3) { float t; asm volatile("v_sub_f32_e64 %0, %1, %2 clamp" : "=v"(t) : "v"(c), "v"(b)); asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(t), "v"(c)); }
                else if (KIND == 4) { float t; asm volatile("v_sub_f32_e32 %0, %1, %2" : "=v"(t) : "v"(a[i]), "v"(b)); a[i] = t; }
                else if (KIND == 5) { float t; asm volatile("v_sub_f32_e64 %0, %1, %2" : "=v"(t) : "v"(a[i]), "v"(b)); a[i] = t; }
            }
    }
    float s = 0; for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, (1 << 22) * 4);
    const char* names[6] = { "v_sub_f32_e64 clamp", "v_fmac_f32_e32", "v_fract_f32", "sub_clamp + fmac (2 instr)", "v_sub_f32_e32", "v_sub_f32_e64" };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int kind = 0; kind < 6; kind++) {
        int iters = 4000;
        dim3 g(256 * 8), b(256);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            switch (kind) {
            case 0: hipLaunchKernelGGL(k<0>, g, b, 0, 0, d, 1.5f, iters); break; case 1: hipLaunchKernelGGL(k<1>, g, b, 0, 0, d, 1.5f, iters); break;
            case 2: hipLaunchKernelGGL(k<2>, g, b, 0, 0, d, 1.5f, iters); break; case 3: hipLaunchKernelGGL(k<3>, g, b, 0, 0, d, 1.5f, iters); break;
            case 4: hipLaunchKernelGGL(k<4>, g, b, 0, 0, d, 1.5f, iters); break; case 5: hipLaunchKernelGGL(k<5>, g, b, 0, 0, d, 1.5f, iters); break; }
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double groups = 8192.0 * iters * 128.0;
        printf("%-28s %.3f ms -> %.3f T groups/s\n", names[kind], ms, groups / (ms * 1e-3) / 1e12);
    }
    return 0;
}
