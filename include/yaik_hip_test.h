/* yaik_hip_test.h — test hooks of libyaik_hip: NOT part of the product ABI.
 *
 * include/yaik_hip.h is the drop-in boundary; the entry points below exist only in the -DYK_TEST_HOOKS build of the same sources
 * (tests/csrc/libyaik_hip_test.so, `make -C yaik_amd/csrc`), which the parity tests load next to the product library: exhaustive
 * self-tests of the arithmetic shortcuts, timing ablations of the fused kernel, and the registration of an independent second
 * implementation of the fused kernel (tests/csrc/yk_encode_v1.hip) to cross-check against. */
#ifndef YAIK_HIP_TEST_H
#define YAIK_HIP_TEST_H
#include "yaik_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- self tests of the arithmetic shortcuts the kernels rely on (exhaustive, run on the device) ----------------
 * which = 0: reciprocal+FMA division == IEEE division for every (minDiff 0..255, value 1..256) pair; *result = mismatches
 * which = 1: floor((n + 0.5) * rcp(scale)) == n / scale for DiffRangeEncode's operands (EncoderContext.cpp:604-623)
 * which = 2: the same shortcut for GetValueModel1's division, 0 <= n < 4096, 1 <= delta <= 255 (EncoderContext.cpp:8383-8391)
 * which = 3: the quantiser table of the fused kernel: for every (min, max) of a tile the LUTs of DynamicTile::buildTable
 *            (EncoderContext.cpp:625-699) equal BN + K[rangeDecode], and every value in [min, max] finds in the table the index
 *            and minDiff the first-minimum scan of GetTileDynamic_Y (:873-881) finds in those LUTs
 * which = 4: the 1-D range kernel's (v * A + B) >> 20 == GetValueModel1's byte for every (delta, minCol, v) (EncoderContext.cpp:8383-8391) */
int yk_selftest(yk_ctx* c, int which, int* result);
/* TIMING ONLY: ablation switches for profiling the fused kernel (results are WRONG while non-zero; default 0).
 * 1 = skip the range quantiser, 2 = skip the gradient passes, 4 = skip the nearest-entry lookups (table gathers; LUT search in
 * the first-generation kernel), 8 = skip the error sums (first-generation kernel).
 * One flag only selects a code path and leaves the results exact (used by the parity tests): 16 = re-sum every tile in the
 * reference's sequential order. */
int yk_set_ablation(yk_ctx* c, int flags);
/* which implementation of the fused kernel yk_encode_tiles launches: 2 (default) = the library's kernel (lane per 4x4 cell); 1 = an
 * external cross-check implementation registered with yk_set_cross_check_launcher (the test suite's first-generation kernel,
 * tests/csrc/yk_encode_v1.hip: lane per pixel row).  Without a registered launcher version 1 fails with YK_ERR_STATE at encode time. */
int yk_set_kernel_version(yk_ctx* c, int version);
/* test hook: fn = int (*)(hipStream_t stream, const YkEncodeParams* params) (yaik_amd/csrc/yk_common.h), launching a kernel that fills the
 * same per-tile outputs from the same parameters; NULL unregisters.  Process-wide. */
int yk_set_cross_check_launcher(void* fn);

#ifdef __cplusplus
}
#endif
#endif
