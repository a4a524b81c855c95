// Library-level entry points: version and the thread-local error string.
#include <algorithm>

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

namespace mphip {
// conv arithmetic policy of the calling thread (mphip_conv3d_set_half_products): a launch reads it when it is issued
static thread_local int g_half_products = 0;
bool conv_half_products() { return g_half_products != 0; }
}  // namespace mphip

extern "C" int mphip_conv3d_set_half_products(int enable) {
    const int prev = mphip::g_half_products;
    mphip::g_half_products = enable ? 1 : 0;
    return prev;
}

extern "C" int mphip_version(void) { return MPHIP_ABI_VERSION; }
// bit 0: some translation unit of this library was compiled with a timing-only ablation on (mphip_ablate.h): its results are wrong by design
extern "C" __attribute__((weak)) int mphip_ablated_build_marker;
extern "C" int mphip_build_flags(void) { return &mphip_ablated_build_marker != nullptr ? 1 : 0; }
extern "C" const char *mphip_last_error(void) { return mphip::g_err; }

namespace mphip {
__global__ void __launch_bounds__(256) zero_fill_kernel(float4 *__restrict__ p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
}  // namespace mphip

namespace mphip {
__global__ void __launch_bounds__(256) absmax_range_kernel(const float *__restrict__ x, size_t n, float *__restrict__ range) {
    unsigned m = 0;
    const size_t n4 = n / 4;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = x4[i];
        m = max(max(m, range_bits(v.x)), max(range_bits(v.y), max(range_bits(v.z), range_bits(v.w))));
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) m = max(m, range_bits(x[i]));
    range_note_block(m, range, blockIdx.x, gridDim.x);
}

int absmax_range_launch(const float *x, size_t n, float *range, hipStream_t s) {
    const size_t per_block = 256 * 4 * 8;  // ~8 float4 per thread
    const unsigned blocks = (unsigned)std::min<size_t>(2048, (n + per_block - 1) / per_block);
    hipLaunchKernelGGL(absmax_range_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, x, n, range);
    return check_launch("absmax_range");
}
}  // namespace mphip

extern "C" int mphip_absmax_range(const float *x, size_t n, float *range, void *stream) {
    MPHIP_REQUIRE(x && range && n > 0, "absmax_range: null pointer or empty tensor");
    MPHIP_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)range & 3) == 0, "absmax_range: x must be 16-byte aligned");
    return mphip::absmax_range_launch(x, n, range, (hipStream_t)stream);
}
