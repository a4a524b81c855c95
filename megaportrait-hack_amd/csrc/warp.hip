// K1/K2/K3 — warp-field composition and the two volumetric warps of the Gbase hot slice.
// HBM-bound kernels: K2 is write-dominated (25 MB out per frame), K3 read-dominated.
// Reference call sites: model.py:965-973/1016-1022 (K1), model.py:1028-1065 (K2),
// model.py:1167-1171 (K3).  Built with -ffp-contract=off: every rounding below is placed where
// ATen's CPU kernels round; the FMAs ATen uses are explicit fmaf().
#include "mphip_common.h"
#include "mphip_resample.h"

namespace mphip {

// ----------------------------------------------------------------------------------------- K1
// One thread per (b,d,h,w); the three components share the index math.
__global__ void __launch_bounds__(256)
warp_field_compose_kernel(const float *__restrict__ theta, const float *__restrict__ em,
                          const float *__restrict__ base, float *__restrict__ wout, float *__restrict__ rt_out,
                          float *__restrict__ em_out, int B, int eD, int eH, int eW, int G) {
    const size_t vol = (size_t)G * G * G;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)B * vol) return;
    int b = (int)(t / vol);
    size_t r = t - (size_t)b * vol;
    int d = (int)(r / ((size_t)G * G));
    int h = (int)((r / G) % G);
    int w = (int)(r % G);
    const float x = base[w], y = base[h], z = base[d];
    const SrcIdx sd = src_index<false>(d, eD, G), sh = src_index<false>(h, eH, G), sw = src_index<false>(w, eW, G);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float *th = theta + ((size_t)b * 3 + j) * 4;
        float acc = x * th[0];
        acc = fmaf(y, th[1], acc);
        acc = fmaf(z, th[2], acc);
        acc = fmaf(1.0f, th[3], acc);
        float e = trilerp(em + ((size_t)b * 3 + j) * eD * eH * eW, eH, eW, sd, sh, sw);
        size_t o = ((size_t)b * 3 + j) * vol + r;
        wout[o] = acc + e;
        if (rt_out) rt_out[o] = acc;
        if (em_out) em_out[o] = e;
    }
}

// ------------------------------------------------------------------------------------ K2 / K3
// Sample coordinate of output voxel (b,d,h,w): the literal op chain of model.py:1036-1058 and
// ATen GridSampler.h:27-36,58-60.  Returns the clipped un-normalised coordinate per axis.
struct Coord3 {
    float x, y, z;
};

__device__ __forceinline__ float coord_axis(float g, float f, float sz) {
    float p = g + f;         // model.py:1052  grid + warp_field
    float m = 2.0f * p;      // model.py:1058  2.0 * warped_grid
    float q = m / sz;        //                / normalization_factors
    float n = q - 1.0f;      //                - 1.0
    float c = ((n + 1.0f) / 2.0f) * sz;  // grid_sampler_unnormalize, align_corners=True
    return fminf(sz, fmaxf(c, 0.0f));    // clip_coordinates (padding_mode='border')
}

__device__ __forceinline__ Coord3 sample_coord(const float *__restrict__ field, const float *__restrict__ lin_d,
                                               const float *__restrict__ lin_h, const float *__restrict__ lin_w,
                                               int b, int d, int h, int w, int D, int H, int W, int fD, int fH,
                                               int fW) {
    const SrcIdx sd = src_index<true>(d, fD, D), sh = src_index<true>(h, fH, H), sw = src_index<true>(w, fW, W);
    const size_t fvol = (size_t)fD * fH * fW;
    const float *fb = field + (size_t)b * 3 * fvol;
    Coord3 c;
    c.x = coord_axis(lin_w[w], trilerp(fb, fH, fW, sd, sh, sw), (float)(W - 1));
    c.y = coord_axis(lin_h[h], trilerp(fb + fvol, fH, fW, sd, sh, sw), (float)(H - 1));
    c.z = coord_axis(lin_d[d], trilerp(fb + 2 * fvol, fH, fW, sd, sh, sw), (float)(D - 1));
    return c;
}

// 8-tap trilinear gather set-up for one voxel: base offset of the (z0,y0,x0) corner, the deltas
// to the +1 corners (0 when that corner is outside: ATen skips it, its weight is 0 there), and
// the 8 corner weights in ATen's accumulation order tnw,tne,tsw,tse,bnw,bne,bsw,bse.
struct Taps {
    int base, dx, dy, dz;
    float w[8];
};

__device__ __forceinline__ Taps make_taps(const Coord3 &c, int D, int H, int W) {
    Taps t;
    int x0 = (int)floorf(c.x), y0 = (int)floorf(c.y), z0 = (int)floorf(c.z);
    float wx1 = c.x - (float)x0, wx0 = (float)(x0 + 1) - c.x;
    float wy1 = c.y - (float)y0, wy0 = (float)(y0 + 1) - c.y;
    float wz1 = c.z - (float)z0, wz0 = (float)(z0 + 1) - c.z;
    bool vx = x0 + 1 < W, vy = y0 + 1 < H, vz = z0 + 1 < D;
    t.base = (z0 * H + y0) * W + x0;
    t.dx = vx ? 1 : 0;
    t.dy = vy ? W : 0;
    t.dz = vz ? H * W : 0;
    if (!vx) wx1 = 0.0f;
    if (!vy) wy1 = 0.0f;
    if (!vz) wz1 = 0.0f;
    t.w[0] = wx0 * wy0 * wz0;
    t.w[1] = wx1 * wy0 * wz0;
    t.w[2] = wx0 * wy1 * wz0;
    t.w[3] = wx1 * wy1 * wz0;
    t.w[4] = wx0 * wy0 * wz1;
    t.w[5] = wx1 * wy0 * wz1;
    t.w[6] = wx0 * wy1 * wz1;
    t.w[7] = wx1 * wy1 * wz1;
    return t;
}

__device__ __forceinline__ float gather8(const float *__restrict__ vol, const Taps &t) {
    const float *p = vol + t.base;
    float acc = 0.0f;
    acc += p[0] * t.w[0];
    acc += p[t.dx] * t.w[1];
    acc += p[t.dy] * t.w[2];
    acc += p[t.dy + t.dx] * t.w[3];
    acc += p[t.dz] * t.w[4];
    acc += p[t.dz + t.dx] * t.w[5];
    acc += p[t.dz + t.dy] * t.w[6];
    acc += p[t.dz + t.dy + t.dx] * t.w[7];
    return acc;
}

// K2: each thread owns VW consecutive w of one (b,d,h) row and a slice of CPB channels
// (blockIdx.y); stores are VW*4-byte vectors, consecutive lanes -> consecutive addresses.
template <int VW>
__global__ void __launch_bounds__(256)
warp_volume_kernel(const float *__restrict__ v, const float *__restrict__ field, const float *__restrict__ lin_d,
                   const float *__restrict__ lin_h, const float *__restrict__ lin_w, float *__restrict__ out,
                   float *__restrict__ coords_out, int32_t *__restrict__ idx_out, int B, int C, int D, int H, int W,
                   int fD, int fH, int fW, int cpb) {
    const int WV = W / VW;
    const size_t nthreads = (size_t)B * D * H * WV;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nthreads) return;
    int wv = (int)(t % WV);
    size_t r = t / WV;
    int h = (int)(r % H);
    r /= H;
    int d = (int)(r % D);
    int b = (int)(r / D);
    const int w0 = wv * VW;
    const size_t vol = (size_t)D * H * W;

    Taps taps[VW];
#pragma unroll
    for (int i = 0; i < VW; ++i) {
        Coord3 c = sample_coord(field, lin_d, lin_h, lin_w, b, d, h, w0 + i, D, H, W, fD, fH, fW);
        taps[i] = make_taps(c, D, H, W);
        if (blockIdx.y == 0 && coords_out) {
            size_t o = ((((size_t)b * D + d) * H + h) * W + w0 + i) * 3;
            coords_out[o] = c.x;
            coords_out[o + 1] = c.y;
            coords_out[o + 2] = c.z;
            if (idx_out) {
                idx_out[o] = (int)floorf(c.x);
                idx_out[o + 1] = (int)floorf(c.y);
                idx_out[o + 2] = (int)floorf(c.z);
            }
        }
    }
    const int c_begin = blockIdx.y * cpb;
    const int c_end = min(C, c_begin + cpb);
    const size_t ooff = ((size_t)d * H + h) * W + w0;
    for (int c = c_begin; c < c_end; ++c) {
        const float *src = v + ((size_t)b * C + c) * vol;
        float res[VW];
#pragma unroll
        for (int i = 0; i < VW; ++i) res[i] = gather8(src, taps[i]);
        float *dst = out + ((size_t)b * C + c) * vol + ooff;
        if (VW == 4) {
            *reinterpret_cast<float4 *>(dst) = make_float4(res[0], res[1], res[2], res[3]);
        } else {
#pragma unroll
            for (int i = 0; i < VW; ++i) dst[i] = res[i];
        }
    }
}

// K3: each thread owns VW consecutive w of one (b,h) row and CPT channels; loops the D output
// slices accumulating the depth projection in registers (sum order d=0..D-1 like torch.sum).
template <int VW, int CPT>
__global__ void __launch_bounds__(256)
warp_volume_dsum_kernel(const float *__restrict__ v, const float *__restrict__ field,
                        const float *__restrict__ lin_d, const float *__restrict__ lin_h,
                        const float *__restrict__ lin_w, float *__restrict__ out, int B, int C, int D, int H, int W,
                        int fD, int fH, int fW) {
    const int WV = W / VW;
    const size_t nthreads = (size_t)B * H * WV;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nthreads) return;
    int wv = (int)(t % WV);
    size_t r = t / WV;
    int h = (int)(r % H);
    int b = (int)(r / H);
    const int w0 = wv * VW;
    const size_t vol = (size_t)D * H * W;
    const int c_begin = blockIdx.y * CPT;

    float sum[CPT][VW];
#pragma unroll
    for (int c = 0; c < CPT; ++c)
#pragma unroll
        for (int i = 0; i < VW; ++i) sum[c][i] = 0.0f;

    for (int d = 0; d < D; ++d) {
        Taps taps[VW];
#pragma unroll
        for (int i = 0; i < VW; ++i) {
            Coord3 c = sample_coord(field, lin_d, lin_h, lin_w, b, d, h, w0 + i, D, H, W, fD, fH, fW);
            taps[i] = make_taps(c, D, H, W);
        }
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            if (c_begin + c < C) {
                const float *src = v + ((size_t)b * C + c_begin + c) * vol;
#pragma unroll
                for (int i = 0; i < VW; ++i) sum[c][i] += gather8(src, taps[i]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        if (c_begin + c < C) {
            float *dst = out + (((size_t)b * C + c_begin + c) * H + h) * W + w0;
            if (VW == 4) {
                *reinterpret_cast<float4 *>(dst) = make_float4(sum[c][0], sum[c][1], sum[c][2], sum[c][3]);
            } else {
#pragma unroll
                for (int i = 0; i < VW; ++i) dst[i] = sum[c][i];
            }
        }
    }
}

}  // namespace mphip

using namespace mphip;

extern "C" int mphip_warp_field_compose(const float *theta, const float *em, const float *base_tbl, float *w,
                                        float *rt_out, float *em_out, int B, int eD, int eH, int eW, int G,
                                        void *stream) {
    MPHIP_REQUIRE(theta && em && base_tbl && w, "warp_field_compose: null pointer");
    MPHIP_REQUIRE(B > 0 && eD > 0 && eH > 0 && eW > 0 && G > 0, "warp_field_compose: bad dims");
    size_t n = (size_t)B * G * G * G;
    hipLaunchKernelGGL(warp_field_compose_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, theta, em,
                       base_tbl, w, rt_out, em_out, B, eD, eH, eW, G);
    return check_launch("warp_field_compose");
}

static int check_warp_args(const char *name, const void *v, const void *field, const void *ld, const void *lh,
                           const void *lw, const void *out, int B, int C, int D, int H, int W, int fD, int fH,
                           int fW) {
    MPHIP_REQUIRE(v && field && ld && lh && lw && out, "%s: null pointer", name);
    MPHIP_REQUIRE(B > 0 && C > 0 && D > 0 && H > 0 && W > 0 && fD > 0 && fH > 0 && fW > 0, "%s: bad dims", name);
    MPHIP_REQUIRE((size_t)D * H * W < (1u << 30), "%s: volume too large for 32-bit tap offsets", name);
    return MPHIP_OK;
}

extern "C" int mphip_warp_volume(const float *v, const float *field, const float *lin_d, const float *lin_h,
                                 const float *lin_w, float *out, float *coords_out, int32_t *idx_out, int B, int C,
                                 int D, int H, int W, int fD, int fH, int fW, void *stream) {
    int rc = check_warp_args("warp_volume", v, field, lin_d, lin_h, lin_w, out, B, C, D, H, W, fD, fH, fW);
    if (rc) return rc;
    MPHIP_REQUIRE(!idx_out || coords_out, "warp_volume: idx_out requires coords_out");
    const int cpb = C >= 48 ? 12 : C;  // channel slice per block row: re-derives coords C/cpb times
    dim3 grid;
    grid.y = cdiv(C, cpb);
    if (W % 4 == 0) {
        grid.x = cdiv((size_t)B * D * H * (W / 4), 256);
        hipLaunchKernelGGL(warp_volume_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, v, field, lin_d, lin_h, lin_w,
                           out, coords_out, idx_out, B, C, D, H, W, fD, fH, fW, cpb);
    } else {
        grid.x = cdiv((size_t)B * D * H * W, 256);
        hipLaunchKernelGGL(warp_volume_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, v, field, lin_d, lin_h, lin_w,
                           out, coords_out, idx_out, B, C, D, H, W, fD, fH, fW, cpb);
    }
    return check_launch("warp_volume");
}

extern "C" int mphip_warp_volume_dsum(const float *v, const float *field, const float *lin_d, const float *lin_h,
                                      const float *lin_w, float *out, int B, int C, int D, int H, int W, int fD,
                                      int fH, int fW, void *stream) {
    int rc = check_warp_args("warp_volume_dsum", v, field, lin_d, lin_h, lin_w, out, B, C, D, H, W, fD, fH, fW);
    if (rc) return rc;
    constexpr int CPT = 8;
    dim3 grid;
    grid.y = cdiv(C, CPT);
    if (W % 4 == 0) {
        grid.x = cdiv((size_t)B * H * (W / 4), 256);
        hipLaunchKernelGGL((warp_volume_dsum_kernel<4, CPT>), grid, dim3(256), 0, (hipStream_t)stream, v, field, lin_d,
                           lin_h, lin_w, out, B, C, D, H, W, fD, fH, fW);
    } else {
        grid.x = cdiv((size_t)B * H * W, 256);
        hipLaunchKernelGGL((warp_volume_dsum_kernel<1, CPT>), grid, dim3(256), 0, (hipStream_t)stream, v, field, lin_d,
                           lin_h, lin_w, out, B, C, D, H, W, fD, fH, fW);
    }
    return check_launch("warp_volume_dsum");
}
