// Library-level entry points: version and the thread-local error string.
#include "mphip_common.h"

namespace mphip {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace mphip

extern "C" int mphip_version(void) { return 1; }
extern "C" const char *mphip_last_error(void) { return mphip::g_err; }

namespace mphip {
__global__ void __launch_bounds__(256) zero_fill_kernel(float4 *__restrict__ p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
}  // namespace mphip
