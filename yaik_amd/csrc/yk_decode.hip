// placeholder until the decode kernels land (next commit)
#include "yk_common.h"
extern "C" {
int yk_decode_begin(yk_ctx* c, int, int) { return yk_fail(c, YK_ERR_STATE, "decode not built yet"); }
int yk_decode_gradient(yk_ctx* c, int, int, const uint8_t*, size_t, const uint8_t*, size_t) { return yk_fail(c, YK_ERR_STATE, "decode not built yet"); }
int yk_decode_1d(yk_ctx* c, const uint8_t*, size_t, const uint8_t*, size_t, int) { return yk_fail(c, YK_ERR_STATE, "decode not built yet"); }
int yk_decode_mask(yk_ctx* c, const uint8_t*, int, int, uint8_t*, size_t) { return yk_fail(c, YK_ERR_STATE, "decode not built yet"); }
int yk_decode_planes(yk_ctx* c, uint8_t*, uint8_t*, uint8_t*, size_t) { return yk_fail(c, YK_ERR_STATE, "decode not built yet"); }
const uint8_t* yk_decode_planes_device(const yk_ctx*, size_t*) { return nullptr; }
int yk_decode_tile4x4(yk_ctx* c, uint8_t*, size_t) { return yk_fail(c, YK_ERR_STATE, "decode not built yet"); }
}
