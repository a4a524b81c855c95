// Shared by the split-f16 ("f16x3") conv translation units (conv3d_f16x3.hip, conv3d_f16x3_wino.hip, mfma_sol.hip).
#pragma once
#include "mphip_common.h"

namespace mphip {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr float F16_CLAMP = 65000.0f;

// v*S = hi + lo, |lo| <= 2^-11 |hi|.  Non-finite or out-of-range values are not clamped: hi takes the value itself
// (Inf / NaN in f16), lo = 0, and the MFMA propagates Inf / NaN like the reference's fp32 conv would.
__device__ __forceinline__ void split_f16(float v, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16)v;                      // |v| > 65504 -> +-Inf, NaN -> NaN
    const float r = v - (float)hi;
    lo = (fabsf(v) <= F16_CLAMP) ? (_Float16)r : (_Float16)0.0f;   // (Inf - Inf would be NaN: keep Inf an Inf)
}

// per-tensor power-of-two weight scale from the packed header's max|w| bits: max|w| * scale < 2^15
__device__ __forceinline__ float weight_scale(unsigned maxbits) {
    float m = __uint_as_float(maxbits);
    if (!(m > 0.0f) || !(m < 1e30f)) return 1.0f;
    int e;
    frexpf(m, &e);              // m = f * 2^e, f in [0.5,1)  ->  m < 2^e
    return ldexpf(1.0f, 15 - e);  // m*scale < 2^15 = 32768
}

// LDS-DMA of 16 bytes per lane (1 KiB per wave): global `src` (per lane) -> LDS byte address `lds` + 16 * lane (wave-uniform
// `lds`).  Issued as inline assembly ON PURPOSE: hipcc's waitcnt pass cannot prove that a later ds_read does not alias the
// destination of __builtin_amdgcn_global_load_lds and puts `s_waitcnt vmcnt(0)` in front of the first LDS read that follows
// it — the wave then sits out the whole L2 round trip (250-400 cycles, MI355X_MICROARCH.md) of a transfer that was meant to
// land an interval later.  The compiler does not see this instruction's vmcnt slot: every consumer waits by hand
// (lds_dma_wait<N>() + a barrier), and the compiler's own `vmcnt(k)` waits stay safe (vmcnt retires in order; an unseen
// younger transfer only makes such a wait longer, never shorter).
__device__ __forceinline__ void lds_dma16(const void *src, unsigned lds) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds) : "memory");   // (m0 is reserved: hipcc re-materialises it before every use of its own)
}
// wait until at most N vector-memory operations of this wave are outstanding (counting the hidden LDS-DMA pieces)
template <int N>
__device__ __forceinline__ void lds_dma_wait() {
    static_assert(N >= 0 && N < 64, "vmcnt is 6 bits on gfx9");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// workgroup barrier that publishes LDS writes and finished LDS reads only (no vmcnt drain: stores and LDS-DMA stay in flight)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace mphip
