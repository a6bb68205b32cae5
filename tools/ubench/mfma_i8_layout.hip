// Checks the operand / result layout of v_mfma_i32_32x32x16_i8 and the semantics of v_permlane32_swap_b32 that yk_lut3d.hip's MFMA scoring relies on.
// hipcc --offload-arch=gfx950 -O2 -o mfma_i8_layout mfma_i8_layout.hip && ./mfma_i8_layout
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ void k(const long* A, const long* B, int* D, unsigned* sw) {
    const int l = threadIdx.x;
    v16i c = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};
    v16i d = __builtin_amdgcn_mfma_i32_32x32x16_i8(A[l], B[l], c, 0, 0, 0);
    for (int v = 0; v < 16; v++) D[l * 16 + v] = d[v];
    auto r = __builtin_amdgcn_permlane32_swap(1000u + l, 2000u + l, false, false);
    sw[l * 2] = r[0]; sw[l * 2 + 1] = r[1];
}
int main() {
    int8_t a[32][16], b[16][32];
    for (int i = 0; i < 32; i++) for (int kk = 0; kk < 16; kk++) a[i][kk] = (int8_t)((i * 7 + kk * 3) % 23 - 11);
    for (int kk = 0; kk < 16; kk++) for (int j = 0; j < 32; j++) b[kk][j] = (int8_t)((j * 5 + kk * 11) % 19 - 9);
    long hA[64], hB[64];
    for (int l = 0; l < 64; l++) {                       // assumed: lane l holds row / column l % 32, k = 8 * (l / 32) .. + 7, byte kk at bits 8 kk
        uint64_t x = 0, y = 0;
        for (int kk = 0; kk < 8; kk++) { x |= (uint64_t)(uint8_t)a[l % 32][8 * (l / 32) + kk] << (8 * kk); y |= (uint64_t)(uint8_t)b[8 * (l / 32) + kk][l % 32] << (8 * kk); }
        hA[l] = (long)x; hB[l] = (long)y;
    }
    long *dA, *dB; int* dD; unsigned* dS;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, 64 * 16 * 4); hipMalloc(&dS, 64 * 2 * 4);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, dS);
    int hD[64 * 16]; unsigned hS[128];
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost); hipMemcpy(hS, dS, sizeof hS, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int v = 0; v < 16; v++) {     // assumed: D[i][j], j = l % 32, i = 8 * (v / 4) + 4 * (l / 32) + v % 4
        const int j = l % 32, i = 8 * (v / 4) + 4 * (l / 32) + (v % 4);
        int ref = 0; for (int kk = 0; kk < 16; kk++) ref += (int)a[i][kk] * (int)b[kk][j];
        if (ref != hD[l * 16 + v]) bad++;
    }
    printf("mfma_i32_32x32x16_i8 layout mismatches: %d of 1024\n", bad);
    int badS = 0;                                        // assumed: result[0] = {old.lo32 lanes, src.lo32 lanes}, result[1] = {old.hi32 lanes, src.hi32 lanes}
    for (int l = 0; l < 64; l++) {
        const unsigned e0 = l < 32 ? 1000u + l : 2000u + (l - 32), e1 = l < 32 ? 1000u + (l + 32) : 2000u + l;
        if (hS[l * 2] != e0 || hS[l * 2 + 1] != e1) badS++;
    }
    printf("permlane32_swap mismatches vs assumption: %d of 64  (lane 0: %u %u, lane 32: %u %u)\n", badS, hS[0], hS[1], hS[64], hS[65]);
    return 0;
}
