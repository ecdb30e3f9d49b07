// placeholder until the backward kernels land (same translation unit will hold them)
#include "vy_common.h"
extern "C" int vy_linear_wgrad(const void*, int64_t, const void*, int64_t, float*, int64_t, float*, float,
                               int64_t, int64_t, int64_t, int, void*) {
  VY_FAIL(VY_ERR_UNSUPPORTED, "vy_linear_wgrad: not built yet");
}
extern "C" int vy_attn_bwd(const void*, int64_t, int64_t, int64_t, const void*, int64_t, int64_t, int64_t,
                           const void*, int64_t, int64_t, int64_t, const void*, const void*, int64_t, int64_t,
                           const float*, float*, void*, int64_t, int64_t, int64_t, void*, int64_t, int64_t,
                           int64_t, void*, int64_t, int64_t, int64_t, int, int64_t, const uint8_t*, int64_t,
                           int64_t, int, int, int64_t, int64_t, int, float, int, void*) {
  VY_FAIL(VY_ERR_UNSUPPORTED, "vy_attn_bwd: not built yet");
}
