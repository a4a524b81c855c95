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

// ---- range descriptors: the per-tensor operand scale of the f16x3 (split-f16) kernels ------------------------------------
// A range descriptor is 4 floats (16 B) on the device:  [0] scale  [1] 1/scale  [2] max|x| (or a rigorous upper bound)  [3] -
//   [0] != 0 : an explicit power-of-two scale (mphip_grad_prep writes gradients' this way);
//   [0] == 0 : the consumer derives the scale from [2]: the power of two with max|x| * scale in [2^13, 2^14).
// Producers (warp gather, GroupNorm apply, ...) zero the descriptor and fold max|out| into [2] with range_note();
// mphip_absmax_range() does it for a tensor of unknown origin.  The f16x3 kernels scale every operand by it before the
// hi/lo split, so no finite value can leave the f16 range (the reference's fp32 conv has no range cliff either) and tensors of
// any magnitude keep fp32-class accuracy; non-finite values are passed through by the split and propagate as Inf/NaN.
__device__ __forceinline__ unsigned range_bits(float v) { return __float_as_uint(fabsf(v)); }  // NaN sorts above Inf above finite

// One wavefront folds its lanes' max (as range_bits) into range[2].  The plain (L1-cached) pre-read skips the atomic once the
// stored maximum already covers this wave — after the first wave of workgroups almost always — so the same-address atomics
// (which serialise in L2) stay a handful per launch.  A stale cached value only causes a redundant atomic, never a missed one.
__device__ __forceinline__ void range_note(unsigned bits, float *__restrict__ range) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) bits = max(bits, (unsigned)__shfl_xor((int)bits, s, 64));
    if ((threadIdx.x & 63) == 0) {
        unsigned *slot = reinterpret_cast<unsigned *>(range) + 2;
        if (bits > *slot) atomicMax(slot, bits);
    }
}

__device__ __forceinline__ void range_scale(const float *__restrict__ range, float &scale, float &inv) {
    const float s0 = range[0];
    if (s0 != 0.0f) {
        scale = s0;
        inv = range[1];
        return;
    }
    const float m = range[2];
    scale = inv = 1.0f;  // all-zero tensor, Inf or NaN inside: unit scale (non-finite values propagate through the split)
    if (m > 0.0f && m < 3.0e38f) {
        int e;
        frexpf(m, &e);  // m = f * 2^e, f in [0.5, 1)  ->  m < 2^e
        e = min(max(14 - e, -120), 120);
        scale = ldexpf(1.0f, e);   // m * scale < 2^14
        inv = ldexpf(1.0f, -e);
    }
}

// max|x| of a tensor of unknown origin -> descriptor (zero-fill + one streaming pass).  Defined in api.hip.
int absmax_range_launch(const float *x, size_t n, float *range, hipStream_t s);

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
