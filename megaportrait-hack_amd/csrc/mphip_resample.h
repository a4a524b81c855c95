// Trilinear / nearest source-index rules shared by the warp and resample kernels (fp32, ATen rules).
#pragma once
#include "mphip_common.h"

namespace mphip {

// ATen UpSample.h area_pixel_compute_source_index + the i0/i1/lambda rule (fp32).
struct SrcIdx {
    int i0, i1;
    float l0, l1;
};

template <bool ALIGN>
__device__ __forceinline__ SrcIdx src_index(int dst, int in, int out) {
    float src;
    if (ALIGN) {
        float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.0f;
        src = scale * (float)dst;
    } else {
        float scale = (float)in / (float)out;
        src = scale * ((float)dst + 0.5f) - 0.5f;
        src = src < 0.0f ? 0.0f : src;
    }
    SrcIdx r;
    r.i0 = min((int)src, in - 1);
    r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
    r.l1 = src - (float)r.i0;
    r.l0 = 1.0f - r.l1;
    return r;
}

__device__ __forceinline__ float lerp2(float l0, float a, float l1, float b) { return fmaf(l0, a, l1 * b); }

// Trilinear sample of one channel plane set `p` ([iD,iH,iW]) — nested W -> H -> D.
__device__ __forceinline__ float trilerp(const float *__restrict__ p, int iH, int iW, const SrcIdx &sd,
                                         const SrcIdx &sh, const SrcIdx &sw) {
    const float *r00 = p + ((size_t)sd.i0 * iH + sh.i0) * iW;
    const float *r01 = p + ((size_t)sd.i0 * iH + sh.i1) * iW;
    const float *r10 = p + ((size_t)sd.i1 * iH + sh.i0) * iW;
    const float *r11 = p + ((size_t)sd.i1 * iH + sh.i1) * iW;
    float a00 = lerp2(sw.l0, r00[sw.i0], sw.l1, r00[sw.i1]);
    float a01 = lerp2(sw.l0, r01[sw.i0], sw.l1, r01[sw.i1]);
    float a10 = lerp2(sw.l0, r10[sw.i0], sw.l1, r10[sw.i1]);
    float a11 = lerp2(sw.l0, r11[sw.i0], sw.l1, r11[sw.i1]);
    float b0 = lerp2(sh.l0, a00, sh.l1, a01);
    float b1 = lerp2(sh.l0, a10, sh.l1, a11);
    return lerp2(sd.l0, b0, sd.l1, b1);
}

}  // namespace mphip
