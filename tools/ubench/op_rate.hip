// micro-benchmark 4: issue rate of the VALU ops the fused kernel could be built from (gfx950); cycles per wave-instruction per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#define OPS(X) \
  X(0,  "v_sub_f32 %0, %0, %1", 1) \
  X(1,  "v_min_f32 %0, %0, %1", 1) \
  X(2,  "v_max_f32 %0, %0, %1", 1) \
  X(3,  "v_min3_f32 %0, %0, %1, %2", 1) \
  X(4,  "v_max3_f32 %0, %0, %1, %2", 1) \
  X(5,  "v_med3_f32 %0, %0, %1, %2", 1) \
  X(6,  "v_min_i32 %0, %0, %1", 0) \
  X(7,  "v_min3_i32 %0, %0, %1, %2", 0) \
  X(8,  "v_pk_min_i16 %0, %0, %1", 0) \
  X(9,  "v_pk_sub_i16 %0, %0, %1 clamp", 0) \
  X(10, "v_pk_mad_i16 %0, %0, %1, %2", 0) \
  X(11, "v_mad_i32_i24 %0, %0, %1, %2", 0) \
  X(12, "v_mul_i32_i24 %0, %0, %1", 0) \
  X(13, "v_and_or_b32 %0, %0, %1, %2", 0) \
  X(14, "v_lshl_or_b32 %0, %0, 1, %2", 0) \
  X(15, "v_perm_b32 %0, %0, %1, %2", 0) \
  X(16, "v_sub_u32 %0, %0, %1", 0) \
  X(17, "v_fma_f32 %0, %0, %1, %2", 1) \
  X(18, "v_mul_f32 %0, %0, %1", 1) \
  X(19, "v_add_f32 %0, |%0|, %1", 1) \
  X(20, "v_cvt_f32_ubyte1 %0, %0", 0) \
  X(21, "v_bfe_u32 %0, %0, 8, 8", 0) \
  X(22, "v_max_f32 %0, |%0|, %1", 1) \
  X(23, "v_pk_max_i16 %0, %0, %1", 0) \
  X(24, "v_cndmask_b32 %0, %0, %1, vcc", 0) \
  X(25, "v_cmp_lt_f32 vcc, %0, %1", 1) \
  X(26, "v_cmp_lt_i32 vcc, %0, %1", 0) \
  X(27, "v_mad_u32_u24 %0, %0, %1, %2", 0) \
  X(28, "v_add3_u32 %0, %0, %1, %2", 0) \
  X(29, "v_mul_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD", 1) \
  X(30, "v_pk_mul_lo_u16 %0, %0, %1", 0) \
  X(31, "v_pk_add_i16 %0, %0, %1", 0) \
  X(32, "v_max_i32 %0, %0, %1", 0) \
  X(33, "v_fmac_f32 %0, %1, %2", 1) \
  X(34, "v_sad_u8 %0, %0, %1, %2", 0) \
  X(35, "v_msad_u8 %0, %0, %1, %2", 0) \
  X(36, "v_sad_u16 %0, %0, %1, %2", 0) \
  X(37, "v_pk_fma_f32 %0, %0, %1, %1", 2) \
  X(38, "v_pk_mul_f32 %0, %0, %1", 2) \
  X(39, "v_pk_add_f32 %0, %0, %1", 2) \
  X(40, "v_pk_mov_b32 %0, %0, %1", 2) \
  X(41, "v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]", 1) \
  X(42, "v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]", 1) \
  X(43, "v_cvt_f32_ubyte2 %0, %0", 0) \
  X(44, "v_cvt_f32_f16 %0, %0", 1) \
  X(45, "v_dot2c_f32_f16 %0, %1, %2", 1) \
  X(46, "v_lshlrev_b64 %0, %1, %0", 3) \
  X(47, "v_lshrrev_b64 %0, %1, %0", 3) \
  X(48, "v_cmp_ne_u64 vcc, %0, %1", 2) \
  X(49, "v_lshl_add_u64 %0, %0, 1, %2", 3) \
  X(50, "v_and_b32 %0, %0, %1", 0) \
  X(51, "v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", 1) \
  X(52, "v_mov_b32_dpp %0, %1 row_shl:4 row_mask:0xf bank_mask:0x5", 0) \
  X(53, "v_alignbit_b32 %0, %0, %1, 4", 0) \
  X(54, "v_add_lshl_u32 %0, %0, %1, 4", 0) \
  X(55, "v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1", 0) \
  X(56, "v_cvt_f32_ubyte0_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2", 0) \
  X(57, "v_dot4_u32_u8 %0, %0, %1, %2", 0) \
  X(58, "v_dot2_u32_u16 %0, %0, %1, %2", 0) \
  X(59, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96", 0)
template <int KIND>
__global__ void k(uint32_t* out, uint32_t seed, int iters) {
    uint32_t a[8]; float f[8]; uint64_t d[8];
    for (int i = 0; i < 8; i++) { a[i] = seed * (threadIdx.x + 1) + i * 0x00010001u; f[i] = 1.0f + (float)(a[i] & 255) * 0.001f; d[i] = ((uint64_t)a[i] << 32) | __float_as_uint(f[i]); }
    uint32_t b = seed * 3 + 0x3c003c00u, c = seed + 0x3c003c00u; float fb = 0.999f, fc = 1.0001f; uint64_t db = ((uint64_t)__float_as_uint(fb) << 32) | __float_as_uint(fc);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
#define X(N, S, F) if (KIND == N) { if (F == 1) asm volatile(S : "+v"(f[i]) : "v"(fb), "v"(fc) : "vcc"); else if (F == 2) asm volatile(S : "+v"(d[i]) : "v"(db), "v"(db) : "vcc"); else if (F == 3) asm volatile(S : "+v"(d[i]) : "v"(b), "v"(db) : "vcc"); else asm volatile(S : "+v"(a[i]) : "v"(b), "v"(c) : "vcc"); }
                OPS(X)
#undef X
            }
    }
    uint32_t s = 0; for (int i = 0; i < 8; i++) s += a[i] + __float_as_uint(f[i]) + (uint32_t)d[i] + (uint32_t)(d[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND> static void run(uint32_t* d, const char* name, hipEvent_t e0, hipEvent_t e1, double clk) {
    const int iters = 2000;
    dim3 g(256 * 8), b(256);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, g, b, 0, 0, d, 3u, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr = 8192.0 * iters * 128.0;          // wave-instructions
    const double rate = instr / (ms * 1e-3);
    printf("%-44.44s %.3f T wave-instr/s  = %.2f cycles per instr per SIMD at %.2f GHz\n", name, rate / 1e12, 1024.0 * clk * 1e9 / rate, clk);
}
int main() {
    uint32_t* d; hipMalloc(&d, (1 << 22) * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double clk = khz / 1e6;
#define X(N, S, F) run<N>(d, S, e0, e1, clk);
    OPS(X)
#undef X
    return 0;
}
