// micro-benchmark 5: how fast does the part start single-wave workgroups?  65536 workgroups of 64 threads that do (almost) nothing,
// with the resources of yk_encode2_kernel (8.4 KB LDS, 128 VGPRs) and without.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDSW, int WORK>
__global__ __launch_bounds__(64, 4) void k(int* out, int n) {
    __shared__ int s[LDSW > 0 ? LDSW : 1];
    int acc = threadIdx.x;
    if (LDSW > 0) { s[threadIdx.x] = acc; __syncthreads(); acc += s[(threadIdx.x + 1) & 63]; }
    for (int i = 0; i < WORK; i++) acc = acc * 1664525 + 1013904223;        // WORK dependent VALU ops
    if (acc == n) out[blockIdx.x] = acc;                                      // never true
}
__global__ __launch_bounds__(64, 4) __attribute__((amdgpu_num_vgpr(128))) void kv(int* out, int n) {
    __shared__ int s[2150];
    int acc = threadIdx.x;
    s[threadIdx.x] = acc; __syncthreads(); acc += s[(threadIdx.x + 1) & 63];
    if (acc == n) out[blockIdx.x] = acc;
}
template <typename F> static void run(const char* name, F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-64s %.1f us for 65536 workgroups = %.0f workgroups per us\n", name, best * 1e3, 65536.0 / (best * 1e3));
}
int main() {
    int* d; hipMalloc(&d, 65536 * 4);
    run("empty, no LDS", [&] { hipLaunchKernelGGL((k<0, 0>), dim3(65536), dim3(64), 0, 0, d, -7); });
    run("empty, 8.4 KB LDS", [&] { hipLaunchKernelGGL((k<2150, 0>), dim3(65536), dim3(64), 0, 0, d, -7); });
    run("empty, 8.4 KB LDS, 128 VGPRs", [&] { hipLaunchKernelGGL(kv, dim3(65536), dim3(64), 0, 0, d, -7); });
    run("2000 dependent VALU ops (~4 us alone), 8.4 KB LDS", [&] { hipLaunchKernelGGL((k<2150, 2000>), dim3(65536), dim3(64), 0, 0, d, -7); });
    run("8000 dependent VALU ops (~16 us alone), 8.4 KB LDS", [&] { hipLaunchKernelGGL((k<2150, 8000>), dim3(65536), dim3(64), 0, 0, d, -7); });
    run("256-thread workgroups x 16384, empty, 33.6 KB LDS", [&] { hipLaunchKernelGGL((k<2150 * 4, 0>), dim3(16384), dim3(256), 0, 0, d, -7); });
    return 0;
}
