// Shared helpers for libmphip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mphip.h"

namespace mphip {

void set_error(const char *fmt, ...);

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return MPHIP_ELAUNCH;
    }
    return MPHIP_OK;
}

#define MPHIP_REQUIRE(cond, ...)          \
    do {                                  \
        if (!(cond)) {                    \
            mphip::set_error(__VA_ARGS__); \
            return MPHIP_EINVAL;          \
        }                                 \
    } while (0)

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// XCD-aware bijective remap of a linear workgroup id: consecutive logical ids land on the same
// XCD (hardware round-robins physical ids over the 8 XCDs), so neighbouring tiles share an L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned nx = 8;
    unsigned q = nwg / nx, r = nwg % nx;
    unsigned xcd = bid % nx, idx = bid / nx;
    unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}


// zero fill as a KERNEL (float4 stores; n_bytes % 16 == 0, 16-B aligned).  Used instead of hipMemsetAsync: the memset
// nodes of a captured hipGraph were not reliably ordered with the kernels around them on ROCm 7.2 (a replayed
// training step went wrong in ~40 % of runs until the two memsets on this path became kernels).
__global__ void __launch_bounds__(256) zero_fill_kernel(float4 *__restrict__ p, size_t n16);
inline void zero_fill(void *p, size_t n_bytes, hipStream_t s) {
    const size_t n16 = n_bytes / 16;
    const unsigned blocks = (unsigned)((n16 + 255) / 256 < 65536 * 16 ? (n16 + 255) / 256 : 65536 * 16);
    hipLaunchKernelGGL(zero_fill_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, (float4 *)p, n16);
}

// value of a split-K tensor element: slab[0][o] + slab[1][o] + ... (z ascending, the reduce kernel's order).
// The loads of 8 slabs are issued together (independent addresses) and only the adds are sequential, so a
// 32-way split costs 4 memory round trips instead of 32.
__device__ __forceinline__ float sum_slabs(const float *__restrict__ x, int splits, size_t slab, size_t o) {
    float v = x[o];
    int z = 1;
    for (; z + 8 <= splits; z += 8) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = x[(size_t)(z + k) * slab + o];
#pragma unroll
        for (int k = 0; k < 8; ++k) v += t[k];
    }
    for (; z < splits; ++z) v += x[(size_t)z * slab + o];
    return v;
}

}  // namespace mphip
