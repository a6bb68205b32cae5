// placeholder until the corner-stream kernels land (next commit)
#include "yk_common.h"
int yk_launch_corners(yk_ctx* c) { return yk_fail(c, YK_ERR_STATE, "corner streams not built yet"); }
extern "C" int yk_gradient_corners(yk_ctx* c, int, uint8_t*, size_t, size_t*) { return yk_fail(c, YK_ERR_STATE, "corner streams not built yet"); }
