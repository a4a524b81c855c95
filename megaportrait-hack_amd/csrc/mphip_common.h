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


// Latency-bound helper kernels (the FlowField chain: small gather convs, one-launch GroupNorms, the dense heads, the field
// compose) raise their wave priority.  When they share CUs with the persistent MFMA workgroups of G3d — the C2D generator on the
// side stream, or a second batch in flight — the issue arbiter otherwise serves the older conv waves first ("priority, then
// age", MI355X_MICROARCH.md) and a 10 us kernel stretches 10-20x.  -DMPHIP_NO_PRIO: same-box A/B builds.
#ifdef MPHIP_NO_PRIO
#define MPHIP_LATENCY_KERNEL_PRIO()
#else
#define MPHIP_LATENCY_KERNEL_PRIO() __builtin_amdgcn_s_setprio(3)
#endif

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
// A range descriptor is MPHIP_RANGE_FLOATS floats on the device (include/mphip.h):
//   [0] scale  [1] 1/scale  [2] max|x| (or a rigorous upper bound)  [3] n (uint)  [4 .. 4+n) per-workgroup partial maxima
//   [0] != 0 : an explicit power-of-two scale (mphip_grad_prep writes gradients' this way);
//   [0] == 0 : the consumer derives the scale from max([2], partials): the power of two with max*scale in [2^13, 2^14).
// Producers (warp gather, GroupNorm apply, mphip_absmax_range, ...) leave ONE partial maximum per workgroup with a plain
// store — no atomics (same-address atomics serialise in one L2 channel: measured +0.27 ms per step), no zero-fill, no
// finalize launch; the consuming conv kernel folds the <= 4096 partials in its prologue (16 KB from L2, once per
// workgroup).  The f16x3 kernels scale every operand by the result before the hi/lo split, so no finite value can leave
// the f16 range (the reference's fp32 conv has no range cliff either) and tensors of any magnitude keep fp32-class
// accuracy; non-finite values are passed through by the split and propagate as Inf/NaN.
constexpr unsigned RANGE_MAX_PARTS = MPHIP_RANGE_FLOATS - 4;

__device__ __forceinline__ unsigned range_bits(float v) { return __float_as_uint(fabsf(v)); }  // NaN sorts above Inf above finite

__device__ __forceinline__ unsigned wave_umax(unsigned v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, s, 64));
    return v;
}

// Producer side.  EVERY thread of the workgroup calls it once, at the end of the kernel (it contains a barrier), with its
// own maximum (range_bits); `block` = linear workgroup index < nblocks <= RANGE_MAX_PARTS.
__device__ __forceinline__ void range_note_block(unsigned bits, float *__restrict__ range, unsigned block, unsigned nblocks) {
#ifdef MPHIP_RANGE_NOTE_OFF  /* dev: isolates the cost of noting ranges */
    return;
#endif
    __shared__ unsigned range_red_[16];
    bits = wave_umax(bits);
    if ((threadIdx.x & 63) == 0) range_red_[threadIdx.x >> 6] = bits;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned m = 0;
        for (unsigned w = 0; w < (blockDim.x + 63) / 64; ++w) m = max(m, range_red_[w]);
        unsigned *r = reinterpret_cast<unsigned *>(range);
        r[4 + block] = m;
        if (block == 0) {
            r[0] = 0;  // derive mode
            r[1] = 0;
            r[2] = 0;
            r[3] = nblocks;
        }
    }
}

__device__ __forceinline__ void scale_from_bits(unsigned bits, float &scale, float &inv) {
    const float m = __uint_as_float(bits);
    scale = inv = 1.0f;  // all-zero tensor, Inf or NaN inside: unit scale (non-finite values propagate through the split)
    if (m > 0.0f && m < 3.0e38f) {
        int e;
        frexpf(m, &e);  // m = f * 2^e, f in [0.5, 1)  ->  m < 2^e
        e = min(max(14 - e, -120), 120);
        scale = ldexpf(1.0f, e);   // m * scale < 2^14
        inv = ldexpf(1.0f, -e);
    }
}

// Consumer side.  EVERY thread of the workgroup calls it (barriers inside); the result is workgroup-uniform.
__device__ __forceinline__ void range_scale_block(const float *__restrict__ range, float &scale, float &inv) {
    if (range[0] != 0.0f) {  // explicit scale (uniform branch)
        scale = range[0];
        inv = range[1];
        return;
    }
    __shared__ unsigned range_fold_[17];
    const unsigned *r = reinterpret_cast<const unsigned *>(range);
    const unsigned n = min(r[3], RANGE_MAX_PARTS);
    unsigned m = r[2];
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x) m = max(m, r[4 + i]);
    m = wave_umax(m);
    if ((threadIdx.x & 63) == 0) range_fold_[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = 0;
        for (unsigned w = 0; w < (blockDim.x + 63) / 64; ++w) t = max(t, range_fold_[w]);
        range_fold_[16] = t;
    }
    __syncthreads();
    scale_from_bits(range_fold_[16], scale, inv);
}

// max|x| of a tensor of unknown origin -> descriptor (one streaming pass, one launch).  Defined in api.hip.
int absmax_range_launch(const float *x, size_t n, float *range, hipStream_t s);

// value of a split-K tensor element: slab[0][o] + slab[1][o] + ... (z ascending, the reduce kernel's order).
// The loads of 8 slabs are issued together (independent addresses) and only the adds are sequential, so a
// 32-way split costs 4 memory round trips instead of 32.
__device__ __forceinline__ float sum_slabs(const float *__restrict__ x, int splits, size_t slab, size_t o) {
    float v = x[o];
    for (int z = 1; z < splits; z += 8) {   // (a short last batch is predicated, not a serial tail: 8 and 16 splits are 1 + 7 and 1 + 8 + 7)
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = z + k < splits ? x[(size_t)(z + k) * slab + o] : 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (z + k < splits) v += t[k];
    }
    return v;
}

}  // namespace mphip
