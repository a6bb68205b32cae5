// yk_roof.hip — the MEASURED HBM roof the roofline figures are quoted against ("denominator = measured
// device-to-device copy bandwidth on the box, reported next to the 8 TB/s specification").  Diagnostics only: no part of the tile path.
//
// Two hand-written streaming kernels, both with 16-byte accesses and several independent loads per lane in flight:
//   * yk_roof_copy_kernel: dst[i] = src[i] (read + write bytes counted) — the figure MI355X_MICROARCH.md quotes (6.29 TB/s, float4 copy);
//   * yk_roof_read_kernel: a read-only stream folded into one word per lane (the fused tile kernel reads 30x more than it writes, so
//     this, not a copy, is the pattern it competes with).
// A framework memcpy (torch's copy_ on uint8: 5.2-5.3 TB/s on this part) understates the roof, which flatters every fraction quoted
// against it; bench.py therefore takes the better of these two kernels as `peak_measured`.
#include "yk_common.h"

typedef uint32_t yk_v4u __attribute__((ext_vector_type(4)));
#define YK_ROOF_THREADS 256
#define YK_ROOF_UNROLL 8

__global__ __launch_bounds__(YK_ROOF_THREADS) void yk_roof_copy_kernel(const yk_v4u* __restrict__ src, yk_v4u* __restrict__ dst, size_t n16) {
    // a workgroup moves contiguous chunks of UNROLL x 4 KB; every lane has UNROLL loads in flight before its first store
    const size_t chunk = (size_t)YK_ROOF_THREADS * YK_ROOF_UNROLL;
    for (size_t base = (size_t)blockIdx.x * chunk; base < n16; base += (size_t)gridDim.x * chunk) {
        yk_v4u v[YK_ROOF_UNROLL];
#pragma unroll
        for (int k = 0; k < YK_ROOF_UNROLL; k++) {
            const size_t i = base + (size_t)k * YK_ROOF_THREADS + threadIdx.x;
            v[k] = i < n16 ? __builtin_nontemporal_load(&src[i]) : (yk_v4u){0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int k = 0; k < YK_ROOF_UNROLL; k++) {
            const size_t i = base + (size_t)k * YK_ROOF_THREADS + threadIdx.x;
            if (i < n16) __builtin_nontemporal_store(v[k], &dst[i]);
        }
    }
}

__global__ __launch_bounds__(YK_ROOF_THREADS) void yk_roof_read_kernel(const yk_v4u* __restrict__ src, size_t n16, uint32_t* __restrict__ sink) {
    const size_t chunk = (size_t)YK_ROOF_THREADS * YK_ROOF_UNROLL;
    uint32_t acc = 0u;
    for (size_t base = (size_t)blockIdx.x * chunk; base < n16; base += (size_t)gridDim.x * chunk) {
        yk_v4u v[YK_ROOF_UNROLL];
#pragma unroll
        for (int k = 0; k < YK_ROOF_UNROLL; k++) {
            const size_t i = base + (size_t)k * YK_ROOF_THREADS + threadIdx.x;
            v[k] = i < n16 ? __builtin_nontemporal_load(&src[i]) : (yk_v4u){0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int k = 0; k < YK_ROOF_UNROLL; k++) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    // the buffer is filled with 0x5A bytes (an even number of equal words per lane xor to 0): the store never happens, the loads cannot be dropped
    if (acc == 0x12345678u) sink[blockIdx.x * YK_ROOF_THREADS + threadIdx.x] = acc;
}

extern "C" int yk_measure_roof(yk_ctx* c, size_t bytes, int reps, double* copyGBs, double* readGBs) {
    if (!c || bytes < (1u << 20) || reps < 1 || reps > 100) return c ? yk_fail(c, YK_ERR_BAD_ARG, "yk_measure_roof arguments") : YK_ERR_BAD_ARG;
    YK_HIP(c, hipSetDevice(c->device));
    bytes &= ~(size_t)4095;
    yk_v4u* a = nullptr; yk_v4u* b = nullptr; uint32_t* sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = YK_OK;
    const size_t n16 = bytes / 16;
    const int grid = c->numCU * 8;                                           // 2048 workgroups of 4 waves: two rounds of full occupancy per sweep step
    auto fail = [&](const char* what, hipError_t e) { rc = yk_fail(c, YK_ERR_HIP, what, e); };
    hipError_t e;
    if ((e = hipMalloc(&a, bytes)) != hipSuccess) fail("hipMalloc roof src", e);
    if (!rc && (e = hipMalloc(&b, bytes)) != hipSuccess) fail("hipMalloc roof dst", e);
    if (!rc && (e = hipMalloc(&sink, (size_t)grid * YK_ROOF_THREADS * 4)) != hipSuccess) fail("hipMalloc roof sink", e);
    if (!rc && (e = hipEventCreate(&e0)) != hipSuccess) fail("hipEventCreate", e);
    if (!rc && (e = hipEventCreate(&e1)) != hipSuccess) fail("hipEventCreate", e);
    if (!rc && (e = hipMemsetAsync(a, 0x5A, bytes, c->stream)) != hipSuccess) fail("hipMemsetAsync", e);
    if (!rc && (e = hipMemsetAsync(b, 0, bytes, c->stream)) != hipSuccess) fail("hipMemsetAsync", e);
    double bestCopy = 0.0, bestRead = 0.0;
    for (int pass = 0; pass < 2 && !rc; pass++) {
        for (int r = 0; r < reps + 1 && !rc; r++) {                          // the first repetition warms up (page tables, clocks)
            if ((e = hipEventRecord(e0, c->stream)) != hipSuccess) { fail("hipEventRecord", e); break; }
            if (pass == 0) hipLaunchKernelGGL(yk_roof_copy_kernel, dim3(grid), dim3(YK_ROOF_THREADS), 0, c->stream, a, b, n16);
            else hipLaunchKernelGGL(yk_roof_read_kernel, dim3(grid), dim3(YK_ROOF_THREADS), 0, c->stream, a, n16, sink);
            if ((e = hipEventRecord(e1, c->stream)) != hipSuccess) { fail("hipEventRecord", e); break; }
            if ((e = hipEventSynchronize(e1)) != hipSuccess) { fail("hipEventSynchronize", e); break; }
            float ms = 0.0f;
            if ((e = hipEventElapsedTime(&ms, e0, e1)) != hipSuccess) { fail("hipEventElapsedTime", e); break; }
            if (r == 0 || ms <= 0.0f) continue;
            const double gbs = (double)(pass == 0 ? 2 * bytes : bytes) / ((double)ms * 1e-3) / 1e9;
            if (pass == 0) bestCopy = gbs > bestCopy ? gbs : bestCopy; else bestRead = gbs > bestRead ? gbs : bestRead;
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (sink) (void)hipFree(sink);
    if (rc) return rc;
    if (copyGBs) *copyGBs = bestCopy;
    if (readGBs) *readGBs = bestRead;
    return YK_OK;
}
