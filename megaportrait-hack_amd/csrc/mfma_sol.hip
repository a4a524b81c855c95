// Measurement only: what rate of the f16x3 convs' MFMA stream does the part SUSTAIN?  (VERDICT r3 #1a)
//
// The conv kernels are priced against the 2.5 PFLOP/s dense f16 peak, which assumes 2.4 GHz; under this instruction mix the package
// sits at its power limit and clocks lower.  This kernel issues the conv's inner stream and nothing else:
// v_mfma_f32_32x32x16_f16 in the three-products-per-tap order (Wlo*Xhi, Whi*Xhi, Whi*Xlo), 3 x 2 accumulator tiles per wave, eight
// waves per workgroup (two per SIMD), one workgroup per CU, on REAL hi/lo fragments (random values split like the conv's operands:
// multiplying zeros draws less power and clocks higher, MI355X_MICROARCH.md "DVFS give-back").  mode 1 also reads its ten 16-byte
// fragments per tap from LDS the way the conv does (same reads per MFMA); mode 0 keeps them in registers.  No global traffic, no
// barriers, no staging: tools/mfma_sol.py / bench.py run it back to back for >= 2 s and report rate, package power and clock —
// the ceiling a conv kernel with perfect overlap could reach on this arithmetic at the power limit.
#include "mphip_f16x3.h"

namespace mphip {

__device__ __forceinline__ unsigned sol_hash(unsigned v) {
    v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16;
    return v;
}

template <int LDS_READS>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
mfma_sol_kernel(float *__restrict__ sink, int iters) {
    constexpr int FRAGS = 64;                                  // 64 hi + 64 lo fragment rows of 1 KiB = 128 KiB
    __shared__ __attribute__((aligned(16))) _Float16 frag[2 * FRAGS * 512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // operands: uniform in (-2^13, 2^13) scaled fp32 values, split hi/lo exactly like the conv's staging
    for (int i = tid; i < FRAGS * 512; i += 512) {
        const unsigned h = sol_hash((unsigned)i * 2654435761u + blockIdx.x * 977u + 12345u);
        const float v = ((float)(h & 0xffffffu) * (1.0f / 8388608.0f) - 1.0f) * 8192.0f;
        _Float16 hi, lo;
        split_f16(v, hi, lo);
        frag[i] = hi;
        frag[FRAGS * 512 + i] = lo;
    }
    __syncthreads();
    f32x16 acc[3][2];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.0f;
    const _Float16 *base = frag + lane * 8 + wave * 512;
    half8 ah[3], al[3], bh[2], bl[2];
#define SOL_LOAD(k_)                                                                                                   \
    {                                                                                                                  \
        _Pragma("unroll") for (int m = 0; m < 3; ++m) {                                                                \
            ah[m] = *reinterpret_cast<const half8 *>(base + (((k_) * 5 + m) & 55) * 512);                              \
            al[m] = *reinterpret_cast<const half8 *>(base + FRAGS * 512 + (((k_) * 5 + m) & 55) * 512);                \
        }                                                                                                              \
        _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                                \
            bh[t] = *reinterpret_cast<const half8 *>(base + (((k_) * 5 + 3 + t) & 55) * 512);                          \
            bl[t] = *reinterpret_cast<const half8 *>(base + FRAGS * 512 + (((k_) * 5 + 3 + t) & 55) * 512);            \
        }                                                                                                              \
    }
    SOL_LOAD(0)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if (LDS_READS) {
                SOL_LOAD(k)
                __builtin_amdgcn_sched_barrier(0);   // (one tap's fragments at a time: hoisting all nine taps' loads spills)
            }
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh[t], acc[m][t], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh[t], acc[m][t], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl[t], acc[m][t], 0, 0, 0);
            if (!LDS_READS) asm volatile("" : "+v"(ah[0]), "+v"(al[0]), "+v"(bh[0]), "+v"(bl[0]));   // (keeps the loop from being folded)
            else __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef SOL_LOAD
    float r = 0.0f;
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) r += acc[m][t][q];
    sink[(size_t)blockIdx.x * 512 + tid] = r;
}

}  // namespace mphip

// sink: >= workgroups * 512 floats.  Issues workgroups * 8 waves * iters * 9 * 18 MFMAs of 32768 FLOP.
extern "C" int mphip_debug_mfma_sol(float *sink, int workgroups, int iters, int mode, void *stream) {
    MPHIP_REQUIRE(sink && workgroups > 0 && iters > 0, "mfma_sol: bad arguments");
    if (mode)
        hipLaunchKernelGGL(mphip::mfma_sol_kernel<1>, dim3(workgroups), dim3(512), 0, (hipStream_t)stream, sink, iters);
    else
        hipLaunchKernelGGL(mphip::mfma_sol_kernel<0>, dim3(workgroups), dim3(512), 0, (hipStream_t)stream, sink, iters);
    return mphip::check_launch("mfma_sol");
}

// Measurement only: the rate at which a workgroup can stream an L2-resident weight tensor into LDS by LDS-DMA the way the F(2,3) conv
// does (8 waves x 3 pieces of 1 KiB per 24 KiB slab, ring of four slabs, a wave waits for the pieces it issued `lag` slabs ago, optional
// barrier per slab).  No MFMAs, no LDS reads: the ceiling of the weight stream alone.
namespace mphip {
template <int LAG, int BARRIER>
__global__ void __launch_bounds__(512) dma_stream_kernel(const _Float16 *__restrict__ w, int slabs_in_tensor, int slabs, float *__restrict__ sink) {
    __shared__ __attribute__((aligned(16))) _Float16 ring[4 * 12288];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) _Float16 *)ring;
    const _Float16 *src0 = w + lane * 8 + wave * 512;
    for (int s = 0; s < slabs; ++s) {
        const _Float16 *src = src0 + (size_t)(s % slabs_in_tensor) * 12288;
        const unsigned dst = lds0 + (unsigned)(s & 3) * 24576u + (unsigned)wave * 1024u;
#pragma unroll
        for (int q = 0; q < 3; ++q) lds_dma16(src + q * 8 * 512, dst + (unsigned)q * 8192u);
        lds_dma_wait<3 * LAG>();
        if (BARRIER) lds_barrier();
    }
    lds_dma_wait<0>();
    __syncthreads();
    sink[blockIdx.x * 512 + threadIdx.x] = (float)ring[threadIdx.x];
}
}  // namespace mphip

extern "C" int mphip_debug_dma_stream(const void *weights, int slabs_in_tensor, int slabs, int lag, int barrier, float *sink, int workgroups, void *stream) {
    MPHIP_REQUIRE(weights && sink && slabs > 0 && slabs_in_tensor > 0 && workgroups > 0, "dma_stream: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const _Float16 *w = (const _Float16 *)weights;
#define DS_LAUNCH(L_, B_) hipLaunchKernelGGL((mphip::dma_stream_kernel<L_, B_>), dim3(workgroups), dim3(512), 0, s, w, slabs_in_tensor, slabs, sink)
    if (lag == 1 && barrier) DS_LAUNCH(1, 1); else if (lag == 2 && barrier) DS_LAUNCH(2, 1); else if (lag == 3 && barrier) DS_LAUNCH(3, 1);
    else if (lag == 1) DS_LAUNCH(1, 0); else if (lag == 2) DS_LAUNCH(2, 0); else DS_LAUNCH(3, 0);
#undef DS_LAUNCH
    return mphip::check_launch("dma_stream");
}
