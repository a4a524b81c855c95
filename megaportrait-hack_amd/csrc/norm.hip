// K6/K7/K8 — GroupNorm (stats + fused apply), resampling and the small head ops.  HBM-bound.
// Reference call sites: nn.GroupNorm model.py:506,508,460,309; AdaptiveGroupNorm 314-316;
// ReLU/residual 517-523, 390-403; AvgPool3d 576-580; nn.Upsample 427-433, 585-589;
// (z+e)@Gamma 945-957; 1x1 Conv2d 446; relu+tanh 462-465.
#include <stdlib.h>

#include "mphip_common.h"
#include "mphip_conv.h"
#include "mphip_resample.h"

namespace mphip {

constexpr int GN_CHUNK = 16384;  // floats reduced by one workgroup (256 threads x 16 float4)

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
    return v;
}

// Stage 1: per (sample,group) span [cnt floats, contiguous in NCDHW] -> per-chunk (sum, sumsq).
// fp32 per-thread partials over <=64 elements, wavefront-shuffle + LDS reduction in double.
__global__ void __launch_bounds__(256)
gn_partial_kernel(const float *__restrict__ x, double *__restrict__ partial, size_t cnt, int chunks) {
    const int grp = blockIdx.y, chunk = blockIdx.x;
    const float *p = x + (size_t)grp * cnt;
    const size_t begin = (size_t)chunk * GN_CHUNK;
    const size_t end = begin + GN_CHUNK < cnt ? begin + GN_CHUNK : cnt;
    float s = 0.0f, ss = 0.0f;
    if ((cnt & 3) == 0 && (((size_t)p) & 15) == 0) {
#pragma unroll 8   // (independent loads: in flight together instead of one round trip per trip; the adds keep their order)
        for (size_t i = begin + (size_t)threadIdx.x * 4; i < end; i += 1024) {
            float4 v = *reinterpret_cast<const float4 *>(p + i);
            s += (v.x + v.y) + (v.z + v.w);
            ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
    } else {
        for (size_t i = begin + threadIdx.x; i < end; i += 256) {
            float v = p[i];
            s += v;
            ss += v * v;
        }
    }
    double ds = wave_sum((double)s), dss = wave_sum((double)ss);
    __shared__ double red[8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        red[wave * 2] = ds;
        red[wave * 2 + 1] = dss;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = (red[0] + red[2]) + (red[4] + red[6]);
        double b = (red[1] + red[3]) + (red[5] + red[7]);
        partial[((size_t)grp * chunks + chunk) * 2] = a;
        partial[((size_t)grp * chunks + chunk) * 2 + 1] = b;
    }
}

// Single-launch variant for groups of <= GN_DIRECT_MAX floats (every tensor below the two largest
// G3d levels): one workgroup walks the whole (sample,group) span and writes (mean, rstd) itself.
constexpr int GN_DIRECT_CHUNKS = 4;

// one (scale, shift) entry of the affine table and its contribution to the bound (gn_affine_table_kernel's arithmetic, shared so that
// the fused and the separate form give the same bits)
__device__ __forceinline__ unsigned gn_table_entry(const GnTable &t, int n, int c, float mean, float rstd) {
    float scale = rstd * t.gamma[c];
    float shift = t.beta[c] - mean * scale;
    float amp = t.gamma[c], off = t.beta[c];
    if (t.w2) {
        scale = scale * t.w2[c];
        shift = shift * t.w2[c] + t.b2[c];
        amp = amp * t.w2[c];
        off = off * t.w2[c] + t.b2[c];
    }
    const int i = n * t.C + c;
    t.table[i * 2] = scale;
    t.table[i * 2 + 1] = shift;
    return range_bits((t.sqrt_ng * fabsf(amp) + fabsf(off)) * 1.0001f);
}

__global__ void __launch_bounds__(256)
gn_stats_direct_kernel(const float *__restrict__ x, float *__restrict__ stats, size_t cnt, float eps, GnTable tbl) {
    const int grp = blockIdx.x;
    const float *p = x + (size_t)grp * cnt;
    double ds = 0.0, dss = 0.0;
    const bool vec = (cnt & 3) == 0 && (((size_t)p) & 15) == 0;
    // (the split-K aware variant is gn_stats_split_kernel below)
    for (size_t begin = 0; begin < cnt; begin += GN_CHUNK) {   // same per-chunk fp32 partials as the 2-stage path
        const size_t end = begin + GN_CHUNK < cnt ? begin + GN_CHUNK : cnt;
        float s = 0.0f, ss = 0.0f;
        if (vec) {
#pragma unroll 8
            for (size_t i = begin + (size_t)threadIdx.x * 4; i < end; i += 1024) {
                float4 v = *reinterpret_cast<const float4 *>(p + i);
                s += (v.x + v.y) + (v.z + v.w);
                ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            }
        } else {
            for (size_t i = begin + threadIdx.x; i < end; i += 256) {
                float v = p[i];
                s += v;
                ss += v * v;
            }
        }
        ds += (double)s;
        dss += (double)ss;
    }
    ds = wave_sum(ds);
    dss = wave_sum(dss);
    __shared__ double red[8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        red[wave * 2] = ds;
        red[wave * 2 + 1] = dss;
    }
    __syncthreads();
    __shared__ float mr_[2];
    if (threadIdx.x == 0) {
        double a = (red[0] + red[2]) + (red[4] + red[6]);
        double b = (red[1] + red[3]) + (red[5] + red[7]);
        double mean = a / (double)cnt;
        double var = b / (double)cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[grp * 2] = mr_[0] = (float)mean;
        stats[grp * 2 + 1] = mr_[1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    if (tbl.table) {   // (workgroup-uniform) this group's channels of the affine table + its share of the range bound
        __syncthreads();
        const int groups = tbl.C / tbl.cpg, n = grp / groups, g = grp % groups;
        unsigned mbits = 0;
        if ((int)threadIdx.x < tbl.cpg) mbits = gn_table_entry(tbl, n, g * tbl.cpg + threadIdx.x, mr_[0], mr_[1]);
        if (tbl.range) range_note_block(mbits, tbl.range, blockIdx.x, gridDim.x);
    }
}

// Statistics of a conv output that is still in split-K form: value(i) = bias[c] + sum_z slab[z][i]
// (z ascending, the order mphip_conv3d_fwd's reduce uses).  Small tensors only: one workgroup per group.
__global__ void __launch_bounds__(1024)
gn_stats_split_kernel(const float *__restrict__ x, int splits, size_t slab, const float *__restrict__ bias,
                      float *__restrict__ stats, int C, int cpg, int S, float eps) {
    const int grp = blockIdx.x;
    const size_t cnt = (size_t)cpg * S, base = (size_t)grp * cnt;
    const int c0 = (int)((base / S) % (size_t)C);
    double ds = 0.0, dss = 0.0;
    {   // flat over the (channel, voxel) span so tiny S (FlowField's 4x1x1 level) still uses every lane
        float s = 0.0f, ss = 0.0f;
#pragma unroll 4
        for (size_t e = threadIdx.x; e < cnt; e += blockDim.x) {
            const size_t o = base + e;
            float v = sum_slabs(x, splits, slab, o);
            if (bias) v += bias[c0 + (int)(e / S)];
            s += v;
            ss += v * v;
        }
        ds = (double)s;
        dss = (double)ss;
    }
    ds = wave_sum(ds);
    dss = wave_sum(dss);
    __shared__ double red[32];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;   // 4 or 16 waves
    if (lane == 0) {
        red[wave * 2] = ds;
        red[wave * 2 + 1] = dss;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < nwaves; w += 4) {   // (groups of four waves, the 4-wave launch's own fold)
            a += (red[w * 2] + red[w * 2 + 2]) + (red[w * 2 + 4] + red[w * 2 + 6]);
            b += (red[w * 2 + 1] + red[w * 2 + 3]) + (red[w * 2 + 5] + red[w * 2 + 7]);
        }
        double mean = a / (double)cnt;
        double var = b / (double)cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[grp * 2] = (float)mean;
        stats[grp * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// Stage 2: (mean, rstd) per (sample,group); biased variance, eps inside the sqrt.
__global__ void gn_finalize_kernel(const double *__restrict__ partial, float *__restrict__ stats, int ngroups,
                                   int chunks, double cnt, float eps, GnTable tbl) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned mbits = 0;
    if (g < ngroups) {
        double s = 0.0, ss = 0.0;
        for (int c = 0; c < chunks; ++c) {
            s += partial[((size_t)g * chunks + c) * 2];
            ss += partial[((size_t)g * chunks + c) * 2 + 1];
        }
        double mean = s / cnt;
        double var = ss / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const float mf = (float)mean, rf = (float)(1.0 / sqrt(var + (double)eps));
        stats[g * 2] = mf;
        stats[g * 2 + 1] = rf;
        if (tbl.table) {
            const int groups = tbl.C / tbl.cpg, n = g / groups, gg = g % groups;
            for (int k = 0; k < tbl.cpg; ++k) mbits = max(mbits, gn_table_entry(tbl, n, gg * tbl.cpg + k, mf, rf));
        }
    }
    if (tbl.table && tbl.range) range_note_block(mbits, tbl.range, blockIdx.x, gridDim.x);   // (every thread of the block arrives)
}

struct GnParams {
    const float *x, *stats, *gamma, *beta, *w2, *b2, *residual;
    float *y;
    int C, cpg;
    int relu, tanh_;
    float *range;  // optional: range descriptor of y (zeroed by the launcher; max|y| is folded in — the next conv's f16x3 scale)
};

__device__ __forceinline__ float gn_value(const GnParams &p, float xv, float mean, float rstd, float g, float b,
                                          float w2, float b2, bool has2, float res, bool has_res) {
    float v = (xv - mean) * rstd * g + b;
    if (has2) v = v * w2 + b2;
    if (has_res) v += res;
    if (p.relu) v = fmaxf(v, 0.0f);
    if (p.tanh_) v = tanhf(v);
    return v;
}

// apply, no pooling: one thread per VW contiguous elements of one (n,c) plane.
template <int VW>
__global__ void __launch_bounds__(256) gn_apply_kernel(GnParams p, int S, size_t total_vec) {
    unsigned mbits = 0;
    // grid-stride: with a range descriptor the grid is capped at the descriptor's RANGE_MAX_PARTS partial-maximum slots
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total_vec; t += (size_t)gridDim.x * blockDim.x) {
        const int SV = S / VW;
        size_t plane = t / SV;  // n*C + c
        int c = (int)(plane % p.C);
        int n = (int)(plane / p.C);
        int grp = n * (p.C / p.cpg) + c / p.cpg;
        float mean = p.stats[grp * 2], rstd = p.stats[grp * 2 + 1];
        float g = p.gamma[c], b = p.beta[c];
        bool has2 = p.w2 != nullptr, has_res = p.residual != nullptr;
        float w2 = has2 ? p.w2[c] : 1.0f, b2 = has2 ? p.b2[c] : 0.0f;
        size_t o = t * VW;
        if (VW == 4) {
            float4 xv = *reinterpret_cast<const float4 *>(p.x + o);
            float4 rv = has_res ? *reinterpret_cast<const float4 *>(p.residual + o) : make_float4(0, 0, 0, 0);
            float4 out;
            out.x = gn_value(p, xv.x, mean, rstd, g, b, w2, b2, has2, rv.x, has_res);
            out.y = gn_value(p, xv.y, mean, rstd, g, b, w2, b2, has2, rv.y, has_res);
            out.z = gn_value(p, xv.z, mean, rstd, g, b, w2, b2, has2, rv.z, has_res);
            out.w = gn_value(p, xv.w, mean, rstd, g, b, w2, b2, has2, rv.w, has_res);
            *reinterpret_cast<float4 *>(p.y + o) = out;
            mbits = max(max(mbits, range_bits(out.x)), max(max(range_bits(out.y), range_bits(out.z)), range_bits(out.w)));
        } else {
            const float v = gn_value(p, p.x[o], mean, rstd, g, b, w2, b2, has2, has_res ? p.residual[o] : 0.0f, has_res);
            p.y[o] = v;
            mbits = max(mbits, range_bits(v));
        }
    }
    if (p.range) range_note_block(mbits, p.range, blockIdx.x, gridDim.x);  // all threads arrive (uniform condition)
}

// apply + AvgPool3d(2,2): one thread per PW pooled output elements along w (sum order kd,kh,kw then /8 like ATen).  PW = 2 (W % 4 == 0):
// 16-byte loads, 8-byte stores, half the index arithmetic per value — the full-resolution launch of G3d's first block (402 MB in, 25 MB
// out at B=8) 90 -> 7x us.
template <int PW>
__global__ void __launch_bounds__(256) gn_apply_pool_kernel(GnParams p, int D, int H, int W, size_t total) {
    unsigned mbits = 0;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const int oD = D / 2, oH = H / 2, oW = W / 2, oWp = oW / PW;
    int ow = (int)(t % oWp) * PW;
    size_t r = t / oWp;
    int oh = (int)(r % oH);
    r /= oH;
    int od = (int)(r % oD);
    size_t plane = r / oD;
    int c = (int)(plane % p.C);
    int n = (int)(plane / p.C);
    int grp = n * (p.C / p.cpg) + c / p.cpg;
    float mean = p.stats[grp * 2], rstd = p.stats[grp * 2 + 1];
    float g = p.gamma[c], b = p.beta[c];
    bool has2 = p.w2 != nullptr, has_res = p.residual != nullptr;
    float w2 = has2 ? p.w2[c] : 1.0f, b2 = has2 ? p.b2[c] : 0.0f;
    float s[PW];
#pragma unroll
    for (int q = 0; q < PW; ++q) s[q] = 0.0f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
            size_t o = ((plane * D + 2 * od + a) * H + 2 * oh + bb) * W + 2 * ow;
            float xv[2 * PW], rv[2 * PW];
            if (PW == 2) {
                const float4 x4 = *reinterpret_cast<const float4 *>(p.x + o);
                xv[0] = x4.x; xv[1] = x4.y; xv[2 * PW - 2] = x4.z; xv[2 * PW - 1] = x4.w;
                const float4 r4 = has_res ? *reinterpret_cast<const float4 *>(p.residual + o) : make_float4(0, 0, 0, 0);
                rv[0] = r4.x; rv[1] = r4.y; rv[2 * PW - 2] = r4.z; rv[2 * PW - 1] = r4.w;
            } else {
                const float2 x2 = *reinterpret_cast<const float2 *>(p.x + o);
                xv[0] = x2.x; xv[1] = x2.y;
                const float2 r2 = has_res ? *reinterpret_cast<const float2 *>(p.residual + o) : make_float2(0, 0);
                rv[0] = r2.x; rv[1] = r2.y;
            }
#pragma unroll
            for (int q = 0; q < PW; ++q) {
                s[q] += gn_value(p, xv[2 * q], mean, rstd, g, b, w2, b2, has2, rv[2 * q], has_res);
                s[q] += gn_value(p, xv[2 * q + 1], mean, rstd, g, b, w2, b2, has2, rv[2 * q + 1], has_res);
            }
        }
    const size_t oo = ((plane * oD + od) * oH + oh) * oW + ow;
    if (PW == 2) *reinterpret_cast<float2 *>(p.y + oo) = make_float2(s[0] / 8.0f, s[PW - 1] / 8.0f);
    else p.y[oo] = s[0] / 8.0f;
#pragma unroll
    for (int q = 0; q < PW; ++q) mbits = max(mbits, range_bits(s[q] / 8.0f));
    }
    if (p.range) range_note_block(mbits, p.range, blockIdx.x, gridDim.x);
}

// General apply for small tensors: x and/or the residual may still be split-K slabs (value = bias[c] +
// sum_z slab[z]), the output may be 2x2x2 average-pooled or nearest-upsampled by (uD,uH,uW) (the
// nn.Upsample that follows FlowField's blocks, model.py:427-433).  One thread per OUTPUT element.
struct GnSplitParams {
    GnParams p;
    int x_splits, res_splits;
    size_t slab;  // elements per slab = N*C*D*H*W
    const float *x_bias, *res_bias;
    int D, H, W, pool2, uD, uH, uW;
};

__device__ __forceinline__ float split_value(const float *__restrict__ x, int splits, size_t slab, size_t o, float bias) {
    return sum_slabs(x, splits, slab, o) + bias;
}

__global__ void __launch_bounds__(256) gn_apply_split_kernel(GnSplitParams q, size_t total) {
    MPHIP_LATENCY_KERNEL_PRIO();
    const GnParams &p = q.p;
    unsigned mbits = 0;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const int oD = q.pool2 ? q.D / 2 : q.D * q.uD, oH = q.pool2 ? q.H / 2 : q.H * q.uH, oW = q.pool2 ? q.W / 2 : q.W * q.uW;
    int ow = (int)(t % oW);
    size_t r = t / oW;
    int oh = (int)(r % oH);
    r /= oH;
    int od = (int)(r % oD);
    size_t plane = r / oD;
    int c = (int)(plane % p.C);
    int n = (int)(plane / p.C);
    int grp = n * (p.C / p.cpg) + c / p.cpg;
    const float mean = p.stats[grp * 2], rstd = p.stats[grp * 2 + 1];
    const float g = p.gamma[c], b = p.beta[c];
    const bool has2 = p.w2 != nullptr, has_res = p.residual != nullptr;
    const float w2 = has2 ? p.w2[c] : 1.0f, b2 = has2 ? p.b2[c] : 0.0f;
    const float xb = (q.x_splits > 1 && q.x_bias) ? q.x_bias[c] : 0.0f;
    const float rb = (q.res_splits > 1 && q.res_bias) ? q.res_bias[c] : 0.0f;
    const size_t pbase = plane * q.D * q.H * q.W;
    if (q.pool2) {
        float s = 0.0f;
        for (int a = 0; a < 2; ++a)
            for (int bb = 0; bb < 2; ++bb)
                for (int cc = 0; cc < 2; ++cc) {
                    size_t o = pbase + ((size_t)(2 * od + a) * q.H + 2 * oh + bb) * q.W + 2 * ow + cc;
                    float xv = split_value(p.x, q.x_splits, q.slab, o, xb);
                    float rv = has_res ? split_value(p.residual, q.res_splits, q.slab, o, rb) : 0.0f;
                    s += gn_value(p, xv, mean, rstd, g, b, w2, b2, has2, rv, has_res);
                }
        p.y[t] = s / 8.0f;
        mbits = max(mbits, range_bits(s / 8.0f));
    } else {
        size_t o = pbase + ((size_t)(od / q.uD) * q.H + oh / q.uH) * q.W + ow / q.uW;
        float xv = split_value(p.x, q.x_splits, q.slab, o, xb);
        float rv = has_res ? split_value(p.residual, q.res_splits, q.slab, o, rb) : 0.0f;
        const float v = gn_value(p, xv, mean, rstd, g, b, w2, b2, has2, rv, has_res);
        p.y[t] = v;
        mbits = max(mbits, range_bits(v));
    }
    }
    if (p.range) range_note_block(mbits, p.range, blockIdx.x, gridDim.x);
}

// Tiny tensors (FlowField: a (sample, group) span of <= GN_FUSED_MAX floats): statistics AND apply in one
// launch, one workgroup per (sample, group).  Pass 1 sums the split-K slabs (+bias) into LDS and reduces
// sum / sum-of-squares; pass 2 normalises from LDS, adds the (possibly split) residual, applies ReLU / tanh and
// writes every nearest-upsampled copy.  Same arithmetic as gn_stats_split_kernel + gn_apply_split_kernel.
constexpr int GN_FUSED_MAX = 12288;  // floats of LDS cache (48 KB)
__global__ void __launch_bounds__(1024) gn_small_fused_kernel(GnSplitParams q, float eps, float *__restrict__ stats_out) {
    MPHIP_LATENCY_KERNEL_PRIO();
    // LDS sized to the span at launch (dynamic): [red: 32 doubles][mr: 2 floats + pad][vals: cnt floats].  A fixed 48 KB cache kept
    // these workgroups from becoming resident beside G3d's persistent conv workgroups (112 KB of the CU's 160 KB), so the C2D
    // generator on the side stream waited for a conv launch to END before each of its GroupNorms could start.
    extern __shared__ __attribute__((aligned(16))) unsigned char gn_dyn_[];
    double *red = reinterpret_cast<double *>(gn_dyn_);
    float *mr = reinterpret_cast<float *>(gn_dyn_ + 256);
    float *vals = reinterpret_cast<float *>(gn_dyn_ + 272);
    const int nthr = blockDim.x, nwaves = blockDim.x >> 6;  // 256 threads, or 1024 when only a few groups exist
    const GnParams &p = q.p;
    const int grp = blockIdx.x;
    const int S = q.D * q.H * q.W;
    const int cnt = p.cpg * S;
    const size_t base = (size_t)grp * cnt;
    const int c0 = (int)((base / S) % (size_t)p.C);
    const int shift = (S & (S - 1)) == 0 ? __ffs(S) - 1 : -1;  // S is a power of two on every hot-path layer
    float s = 0.0f, ss = 0.0f;
#pragma unroll 4
    for (int e = threadIdx.x; e < cnt; e += nthr) {
        const int c = shift >= 0 ? (e >> shift) : e / S;
        float v = sum_slabs(p.x, q.x_splits, q.slab, base + e);
        if (q.x_splits > 1 && q.x_bias) v += q.x_bias[c0 + c];
        vals[e] = v;
        s += v;
        ss += v * v;
    }
    double ds = wave_sum((double)s), dss = wave_sum((double)ss);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        red[wave * 2] = ds;
        red[wave * 2 + 1] = dss;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int wv = 0; wv < nwaves; ++wv) {
            a += red[wv * 2];
            b += red[wv * 2 + 1];
        }
        double mean = a / (double)cnt;
        double var = b / (double)cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        mr[0] = (float)mean;
        mr[1] = (float)(1.0 / sqrt(var + (double)eps));
        if (stats_out) {
            stats_out[grp * 2] = mr[0];
            stats_out[grp * 2 + 1] = mr[1];
        }
    }
    __syncthreads();
    const float mean = mr[0], rstd = mr[1];
    const bool has2 = p.w2 != nullptr, has_res = p.residual != nullptr;
    const int HW = q.H * q.W;
    const int oH = q.H * q.uH, oW = q.W * q.uW, oD = q.D * q.uD;
    const size_t oS = (size_t)oD * oH * oW;
    const int n_c0 = (int)(base / S);  // global (n*C + c) index of the group's first channel plane
    unsigned mbits = 0;
#pragma unroll 4
    for (int e = threadIdx.x; e < cnt; e += nthr) {
        const int c = shift >= 0 ? (e >> shift) : e / S;
        const int i = e - c * S;
        const int ch = c0 + c;
        float rv = 0.0f;
        if (has_res) {
            rv = sum_slabs(p.residual, q.res_splits, q.slab, base + e);
            if (q.res_splits > 1 && q.res_bias) rv += q.res_bias[ch];
        }
        const float v = gn_value(p, vals[e], mean, rstd, p.gamma[ch], p.beta[ch], has2 ? p.w2[ch] : 1.0f,
                                 has2 ? p.b2[ch] : 0.0f, has2, rv, has_res);
        mbits = max(mbits, range_bits(v));
        const int d = i / HW, h = (i / q.W) % q.H, w = i % q.W;
        float *dst = p.y + (size_t)(n_c0 + c) * oS;
        for (int a = 0; a < q.uD; ++a)
            for (int b = 0; b < q.uH; ++b)
                for (int cc = 0; cc < q.uW; ++cc)
                    dst[((size_t)(d * q.uD + a) * oH + h * q.uH + b) * oW + w * q.uW + cc] = v;
    }
    if (p.range) range_note_block(mbits, p.range, blockIdx.x, gridDim.x);
}

__global__ void __launch_bounds__(256) avgpool2_kernel(const float *__restrict__ x, float *__restrict__ y, int D,
                                                       int H, int W, size_t total) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int oD = D / 2, oH = H / 2, oW = W / 2;
    int ow = (int)(t % oW);
    size_t r = t / oW;
    int oh = (int)(r % oH);
    r /= oH;
    int od = (int)(r % oD);
    size_t plane = r / oD;
    float s = 0.0f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float *q = x + ((plane * D + 2 * od + a) * H + 2 * oh + b) * W + 2 * ow;
            s += q[0];
            s += q[1];
        }
    y[t] = s / 8.0f;
}

// nn.Upsample(scale_factor=2, trilinear, align_corners=True): src = dst*scale, scale = (in-1)/(out-1)
// evaluated once in fp32 on the host (the same single division ATen performs).  One thread makes 4
// consecutive outputs along w (one 16-byte store); the d/h source rows are shared by the four.
__device__ __forceinline__ SrcIdx src_index_scaled(int dst, int in, float scale) {
    float src = scale * (float)dst;
    SrcIdx r;
    r.i0 = min((int)src, in - 1);
    r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
    r.l1 = src - (float)r.i0;
    r.l0 = 1.0f - r.l1;
    return r;
}

// One thread makes a 2(d) x 2(h) x 4(w) brick of outputs (four 16-byte stores): the brick's sources are a
// 3 x 3 x 4 neighbourhood at most, read once into registers (36 loads for 16 outputs instead of 128), and
// every output is the same nested W->H->D lerp as the scalar form, so results stay bit-identical.
__global__ void __launch_bounds__(256)
upsample_trilinear2_kernel(const float *__restrict__ x, float *__restrict__ y, int D, int H, int W, float sD, float sH,
                           float sW, size_t nbricks) {
  {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nbricks) return;
    const int oH = 2 * H, oW = 2 * W;
    const int bw = W / 2;  // bricks per output row (2W / 4)
    int kw = (int)(t % bw);
    size_t r = t / bw;
    int kh = (int)(r % H);      // output rows 2kh, 2kh+1
    r /= H;
    int kd = (int)(r % D);      // output slices 2kd, 2kd+1
    size_t plane = r / D;
    const float *p = x + plane * D * H * W;
    SrcIdx sd[2], sh[2], sw[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        sd[i] = src_index_scaled(2 * kd + i, D, sD);
        sh[i] = src_index_scaled(2 * kh + i, H, sH);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) sw[i] = src_index_scaled(4 * kw + i, W, sW);
    // distinct source indices: d in {d_lo..d_lo+2}, h likewise, w in {w_lo..w_lo+3}
    const int d_lo = sd[0].i0, h_lo = sh[0].i0, w_lo = sw[0].i0;
    float v[3][3][4];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const float *row = p + ((size_t)min(d_lo + a, D - 1) * H + min(h_lo + b, H - 1)) * W;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[a][b][c] = row[min(w_lo + c, W - 1)];
        }
    // separable evaluation, identical operand pairs to the nested form: W, then H, then D
    // (sequential selects: the nested-ternary form compiled into ~120 divergent branch ladders — s_and_saveexec / s_cbranch per select —
    //  instead of v_cndmask chains; r04)
    auto sel3 = [](int k, float q0, float q1, float q2) -> float {
        float r = q2;
        r = k == 1 ? q1 : r;
        r = k == 0 ? q0 : r;
        return r;
    };
    auto sel4 = [](int k, float q0, float q1, float q2, float q3) -> float {
        float r = q3;
        r = k == 2 ? q2 : r;
        r = k == 1 ? q1 : r;
        r = k == 0 ? q0 : r;
        return r;
    };
    float wl[3][3][4];  // [src d][src h][out w]
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int c0 = sw[c].i0 - w_lo, c1 = sw[c].i1 - w_lo;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
                wl[a][b][c] = lerp2(sw[c].l0, sel4(c0, v[a][b][0], v[a][b][1], v[a][b][2], v[a][b][3]), sw[c].l1,
                                    sel4(c1, v[a][b][0], v[a][b][1], v[a][b][2], v[a][b][3]));
    }
    float hl[3][2][4];  // [src d][out h][out w]
#pragma unroll
    for (int jh = 0; jh < 2; ++jh) {
        const int b0 = sh[jh].i0 - h_lo, b1 = sh[jh].i1 - h_lo;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                hl[a][jh][c] = lerp2(sh[jh].l0, sel3(b0, wl[a][0][c], wl[a][1][c], wl[a][2][c]), sh[jh].l1,
                                     sel3(b1, wl[a][0][c], wl[a][1][c], wl[a][2][c]));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int a0 = sd[i].i0 - d_lo, a1 = sd[i].i1 - d_lo;
#pragma unroll
        for (int jh = 0; jh < 2; ++jh) {
            float o[4];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                o[c] = lerp2(sd[i].l0, sel3(a0, hl[0][jh][c], hl[1][jh][c], hl[2][jh][c]), sd[i].l1,
                             sel3(a1, hl[0][jh][c], hl[1][jh][c], hl[2][jh][c]));
            float *dst = y + ((plane * 2 * D + 2 * kd + i) * oH + 2 * kh + jh) * (size_t)oW + 4 * kw;
            *reinterpret_cast<float4 *>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
  }
}

// Demand-driven x2 upsample (mphip_upsample_trilinear2_roi): one workgroup per channel plane walks ONLY the bricks inside the
// region its frame's sample box needs (the box's conv tiles grown by the 1-voxel conv halo) — a ~5^3 box is ~100 of a plane's
// 4096 bricks.  (The first version tested every brick of the full launch: 47-74 us of dispatch and index math for 1 % useful work.)
__global__ void __launch_bounds__(256)
upsample_trilinear2_roi_kernel(const float *__restrict__ x, float *__restrict__ y, int D, int H, int W, float sD, float sH, float sW,
                               const int *__restrict__ roi, int roi_frames, int C, int gD, int gH, int gW) {
    const size_t plane = blockIdx.x;
    const int first = roi_frames > 0 ? 0 : (int)(plane / C), count = roi_frames > 0 ? roi_frames : 1;
    // needed output region of this plane: the union of the boxes' regions (its bounding region: a superset is harmless)
    int wl = 1 << 30, wh = -1, hl = 1 << 30, hh = -1, dl = 1 << 30, dh = -1;
    for (int f = first; f < first + count; ++f) {
        const int *b = roi + f * 8;
        wl = min(wl, (b[0] / gW) * gW - 1); wh = max(wh, ((b[0] + b[3] - 1) / gW + 1) * gW);
        hl = min(hl, (b[1] / gH) * gH - 1); hh = max(hh, ((b[1] + b[4] - 1) / gH + 1) * gH);
        dl = min(dl, (b[2] / gD) * gD - 1); dh = max(dh, ((b[2] + b[5] - 1) / gD + 1) * gD);
    }
    // -> brick ranges (a brick = output slices 2kd..2kd+1, rows 2kh..2kh+1, columns 4kw..4kw+3)
    const int kw0 = max(wl, 0) / 4, kw1 = min(wh, 2 * W - 1) / 4, kh0 = max(hl, 0) / 2, kh1 = min(hh, 2 * H - 1) / 2;
    const int kd0 = max(dl, 0) / 2, kd1 = min(dh, 2 * D - 1) / 2;
    const int nw = kw1 - kw0 + 1, nh = kh1 - kh0 + 1, nd = kd1 - kd0 + 1;
    if (nw <= 0 || nh <= 0 || nd <= 0) return;
    const float *p = x + plane * D * H * W;
    const int oH = 2 * H, oW = 2 * W;
    for (int t = threadIdx.x; t < nw * nh * nd; t += 256) {
        const int kw = kw0 + t % nw, kh = kh0 + (t / nw) % nh, kd = kd0 + t / (nw * nh);
        SrcIdx sd[2], sh[2], sw[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            sd[i] = src_index_scaled(2 * kd + i, D, sD);
            sh[i] = src_index_scaled(2 * kh + i, H, sH);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) sw[i] = src_index_scaled(4 * kw + i, W, sW);
        // one output at a time, the nested W -> H -> D lerp of the scalar form (bit-identical to the brick kernel's separable
        // evaluation: the same operand pairs in the same order)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jh = 0; jh < 2; ++jh) {
                float o[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float hv[2][2];
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b2 = 0; b2 < 2; ++b2) {
                            const int sdi = a ? sd[i].i1 : sd[i].i0, shi = b2 ? sh[jh].i1 : sh[jh].i0;
                            const float *row = p + ((size_t)sdi * H + shi) * W;
                            hv[a][b2] = lerp2(sw[c].l0, row[sw[c].i0], sw[c].l1, row[sw[c].i1]);
                        }
                    const float d0v = lerp2(sh[jh].l0, hv[0][0], sh[jh].l1, hv[0][1]);
                    const float d1v = lerp2(sh[jh].l0, hv[1][0], sh[jh].l1, hv[1][1]);
                    o[c] = lerp2(sd[i].l0, d0v, sd[i].l1, d1v);
                }
                float *dst = y + ((plane * 2 * D + 2 * kd + i) * oH + 2 * kh + jh) * (size_t)oW + 4 * kw;
                *reinterpret_cast<float4 *>(dst) = make_float4(o[0], o[1], o[2], o[3]);
            }
    }
}

// generic (odd W) fallback: one output per thread
__global__ void __launch_bounds__(256) upsample_trilinear2_scalar_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                                         int D, int H, int W, size_t total) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int oD = 2 * D, oH = 2 * H, oW = 2 * W;
    int ow = (int)(t % oW);
    size_t r = t / oW;
    int oh = (int)(r % oH);
    r /= oH;
    int od = (int)(r % oD);
    size_t plane = r / oD;
    SrcIdx sd = src_index<true>(od, D, oD), sh = src_index<true>(oh, H, oH), sw = src_index<true>(ow, W, oW);
    y[t] = trilerp(x + plane * D * H * W, H, W, sd, sh, sw);
}

// F.interpolate(x, scale_factor=(sD,sH,sW), mode='trilinear', align_corners=False) for integer scale factors — the
// `upsample=True` branch of ResBlock3D / ResBlock3D_Adaptive (model.py:404-405, 525-526; no module of Gbase sets it).
__global__ void __launch_bounds__(256) upsample_trilinear_scaled_kernel(const float *__restrict__ x, float *__restrict__ y, int D,
                                                                        int H, int W, int sD, int sH, int sW, size_t total) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int oD = D * sD, oH = H * sH, oW = W * sW;
    int ow = (int)(t % oW);
    size_t r = t / oW;
    int oh = (int)(r % oH);
    r /= oH;
    int od = (int)(r % oD);
    size_t plane = r / oD;
    SrcIdx sd = src_index<false>(od, D, oD), sh = src_index<false>(oh, H, oH), sw = src_index<false>(ow, W, oW);
    y[t] = trilerp(x + plane * D * H * W, H, W, sd, sh, sw);
}

__global__ void __launch_bounds__(256) upsample_nearest_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                               int D, int H, int W, int sD, int sH, int sW,
                                                               size_t total) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int oD = D * sD, oH = H * sH, oW = W * sW;
    int ow = (int)(t % oW);
    size_t r = t / oW;
    int oh = (int)(r % oH);
    r /= oH;
    int od = (int)(r % oD);
    size_t plane = r / oD;
    y[t] = x[((plane * D + od / sD) * H + oh / sH) * W + ow / sW];
}

// out[b,n] = sum_k (a[b,k]+a2[b,k]) * M(k,n) + bias[n];  M(k,n) = m[k*N+n] (trans=0) or m[n*K+k] (trans=1).
// trans=1 (row-major [N][K], the Conv2d weight layout): one wave per output column n, lanes stride over k.
template <bool TRANS>
__global__ void __launch_bounds__(256)
add_matmul_kernel(const float *__restrict__ a, const float *__restrict__ a2, const float *__restrict__ m,
                  const float *__restrict__ bias, float *__restrict__ out, int B, int K, int N) {
    MPHIP_LATENCY_KERNEL_PRIO();
    constexpr int BMAX = 8;
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    for (int b0 = 0; b0 < B; b0 += BMAX) {
        float acc[BMAX];
#pragma unroll
        for (int i = 0; i < BMAX; ++i) acc[i] = 0.0f;
        for (int k = lane; k < K; k += 64) {
            float mv = TRANS ? m[(size_t)n * K + k] : m[(size_t)k * N + n];
#pragma unroll
            for (int i = 0; i < BMAX; ++i) {
                if (b0 + i < B) {
                    float av = a[(size_t)(b0 + i) * K + k];
                    if (a2) av += a2[(size_t)(b0 + i) * K + k];
                    acc[i] = fmaf(av, mv, acc[i]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < BMAX; ++i) {
            float v = acc[i];
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
            if (lane == 0 && b0 + i < B) out[(size_t)(b0 + i) * N + n] = v + (bias ? bias[n] : 0.0f);
        }
    }
}

// trans=0 fast path ([K][N] matrix, N contiguous): a workgroup of 8 waves owns 16 output columns; a wave-iteration reads four
// 64-byte row segments (lane = 16*r + col), so K = 512 is 16 iterations per wave and N = 2048 gives 128 workgroups (the 64-column
// version ran the generators' 4 MB head product on 32 workgroups: 35 us, 0.12 TB/s).  The (a+a2) rows are broadcast from LDS; the
// 4 x 8 partial sums per output are combined by two shuffles and an LDS pass in wave order (deterministic).
// (8 waves, 16 KB + 4 KB of LDS: also resident beside a 112 KB conv workgroup — see gn_small_fused_kernel)
constexpr int MM_WAVES = 8, MM_BMAX = 8, MM_KMAX = 512, MM_COLS = 16;
__global__ void __launch_bounds__(MM_WAVES * 64)
add_matmul_kn_kernel(const float *__restrict__ a, const float *__restrict__ a2, const float *__restrict__ m,
                     const float *__restrict__ bias, float *__restrict__ out, int B, int K, int N, int b0) {
    MPHIP_LATENCY_KERNEL_PRIO();
    __shared__ float as[MM_BMAX * MM_KMAX];
    __shared__ float part[MM_WAVES][MM_BMAX][MM_COLS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & (MM_COLS - 1), r = lane / MM_COLS;   // 4 rows per wave-iteration
    const int nb = min(MM_BMAX, B - b0);
    for (int i = threadIdx.x; i < nb * K; i += MM_WAVES * 64) {
        const int bb = i / K, k = i - bb * K;
        float v = a[(size_t)(b0 + bb) * K + k];
        if (a2) v += a2[(size_t)(b0 + bb) * K + k];
        as[bb * MM_KMAX + k] = v;
    }
    __syncthreads();
    const int n = blockIdx.x * MM_COLS + col;
    float acc[MM_BMAX];
#pragma unroll
    for (int i = 0; i < MM_BMAX; ++i) acc[i] = 0.0f;
    if (n < N) {
        constexpr int RPI = MM_WAVES * (64 / MM_COLS);   // rows per workgroup-iteration
#pragma unroll 4
        for (int k = wave * (64 / MM_COLS) + r; k < K; k += RPI) {
            const float mv = m[(size_t)k * N + n];
#pragma unroll
            for (int i = 0; i < MM_BMAX; ++i) acc[i] = fmaf(as[i * MM_KMAX + k], mv, acc[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < MM_BMAX; ++i) {
        float v = acc[i];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (r == 0) part[wave][i][col] = v;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < nb * MM_COLS; o += MM_WAVES * 64) {
        const int bb = o / MM_COLS, l = o % MM_COLS;
        const int nn = blockIdx.x * MM_COLS + l;
        if (nn >= N) continue;
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < MM_WAVES; ++w) v += part[w][bb][l];
        out[(size_t)(b0 + bb) * N + nn] = v + (bias ? bias[nn] : 0.0f);
    }
}

// K0: theta[b] = rows 0..2 of A = [R|t; 0 0 0 1], optionally inverted (Gauss-Jordan, partial pivoting).
__global__ void rt_theta_kernel(const float *__restrict__ rot, const float *__restrict__ tr, float *__restrict__ theta,
                                int B, int invert) {
    MPHIP_LATENCY_KERNEL_PRIO();
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float k = 0.017453292519943295f;  // torch.pi / 180.0 as fp32 (model.py:823)
    float ra = rot[b * 3] * k, rb = rot[b * 3 + 1] * k, rg = rot[b * 3 + 2] * k;
    float ca = cosf(ra), sa = sinf(ra), cb = cosf(rb), sb = sinf(rb), cg = cosf(rg), sg = sinf(rg);
    float Rx[3][3] = {{1, 0, 0}, {0, ca, -sa}, {0, sa, ca}};
    float Ry[3][3] = {{cb, 0, sb}, {0, 1, 0}, {-sb, 0, cb}};
    float Rz[3][3] = {{cg, -sg, 0}, {sg, cg, 0}, {0, 0, 1}};
    float T[3][3], R[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float s = 0.0f;
            for (int q = 0; q < 3; ++q) s = fmaf(Ry[i][q], Rz[q][j], s);
            T[i][j] = s;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float s = 0.0f;
            for (int q = 0; q < 3; ++q) s = fmaf(Rx[i][q], T[q][j], s);
            R[i][j] = s;
        }
    float A[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            A[i][j] = i < 3 ? (j < 3 ? R[i][j] : tr[b * 3 + i]) : (j == 3 ? 1.0f : 0.0f);
            A[i][4 + j] = i == j ? 1.0f : 0.0f;
        }
    if (invert) {
        for (int col = 0; col < 4; ++col) {
            int piv = col;
            float best = fabsf(A[col][col]);
            for (int r = col + 1; r < 4; ++r)
                if (fabsf(A[r][col]) > best) { best = fabsf(A[r][col]); piv = r; }
            if (piv != col)
                for (int j = 0; j < 8; ++j) { float tmp = A[col][j]; A[col][j] = A[piv][j]; A[piv][j] = tmp; }
            float inv = 1.0f / A[col][col];
            for (int j = 0; j < 8; ++j) A[col][j] *= inv;
            for (int r = 0; r < 4; ++r) {
                if (r == col) continue;
                float f = A[r][col];
                for (int j = 0; j < 8; ++j) A[r][j] = fmaf(-f, A[col][j], A[r][j]);
            }
        }
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) theta[(b * 3 + i) * 4 + j] = invert ? A[i][4 + j] : A[i][j];
}

}  // namespace mphip

using namespace mphip;

extern "C" int mphip_rt_theta(const float *rot, const float *tr, float *theta, int B, int invert, void *stream) {
    MPHIP_REQUIRE(rot && tr && theta, "rt_theta: null pointer");
    MPHIP_REQUIRE(B > 0, "rt_theta: bad batch");
    hipLaunchKernelGGL(rt_theta_kernel, dim3(cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, rot, tr, theta, B, invert);
    return check_launch("rt_theta");
}

__global__ void gn_affine_table_kernel(const float *__restrict__ stats, const float *__restrict__ gamma, const float *__restrict__ beta,
                                       const float *__restrict__ w2, const float *__restrict__ b2, float *__restrict__ table,
                                       float *__restrict__ range, int N, int C, int cpg, float sqrt_ng);

namespace mphip {
// Statistics of a conv output from the conv's own epilogue partials (conv3d_f16x3.hip): part[C][tile][wave][2] holds (sum, sum of
// squares) of the raw accumulators ((value - bias) / u, with u stored behind the partials) over the 64-or-so voxels a wave owned, in
// fp32; a frame's tiles are consecutive.  One workgroup per (sample, group) folds tiles x waves x channels-of-the-group in double and
// writes (mean, rstd) (+ the affine table entries, as gn_stats_direct_kernel).
__global__ void __launch_bounds__(256)
gn_tile_finalize_kernel(const float *__restrict__ part, const float *__restrict__ bias, float *__restrict__ stats, int tiles_per_frame,
                        int waves, int C, int cpg, double cnt, double vox_per_row, float eps, GnTable tbl) {
    const int grp = blockIdx.x, groups = C / cpg, n = grp / groups, g = grp % groups;
    const int rows = tiles_per_frame * waves;   // (tile, wave) rows of this frame
    const size_t all_rows = (size_t)(gridDim.x / groups) * rows;
    const double u = (double)part[(size_t)C * all_rows * 2];   // the conv's accumulator -> value factor, stored behind the partials
    double ds = 0.0, dss = 0.0;
    for (int c = 0; c < cpg; ++c) {
        const float2 *p = reinterpret_cast<const float2 *>(part) + (size_t)(g * cpg + c) * all_rows + (size_t)n * rows;
        const double b = bias ? (double)bias[g * cpg + c] : 0.0;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll 4
        for (int r = threadIdx.x; r < rows; r += 256) {   // sums of the raw accumulators ((value - bias[c]) / u) over a row's voxels
            const float2 v = p[r];
            s1 += (double)v.x;
            s2 += (double)v.y;
        }
        const double nr = (double)((rows - (int)threadIdx.x + 255) / 256) * vox_per_row;   // voxels behind this thread's rows
        const double sa = u * s1;
        ds += sa + nr * b;
        dss += u * u * s2 + 2.0 * b * sa + nr * b * b;
    }
    ds = wave_sum(ds);
    dss = wave_sum(dss);
    __shared__ double red[8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        red[wave * 2] = ds;
        red[wave * 2 + 1] = dss;
    }
    __syncthreads();
    __shared__ float mr_[2];
    if (threadIdx.x == 0) {
        double a = (red[0] + red[2]) + (red[4] + red[6]);
        double b = (red[1] + red[3]) + (red[5] + red[7]);
        double mean = a / cnt;
        double var = b / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[grp * 2] = mr_[0] = (float)mean;
        stats[grp * 2 + 1] = mr_[1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    if (tbl.table) {
        __syncthreads();
        unsigned mbits = 0;
        if ((int)threadIdx.x < tbl.cpg) mbits = gn_table_entry(tbl, n, g * tbl.cpg + threadIdx.x, mr_[0], mr_[1]);
        if (tbl.range) range_note_block(mbits, tbl.range, blockIdx.x, gridDim.x);
    }
}

size_t groupnorm_ws_bytes(int N, int C, int S, int G) {
    if (N <= 0 || C <= 0 || S <= 0 || G <= 0 || C % G) return 0;
    size_t cnt = (size_t)(C / G) * S;
    size_t chunks = (cnt + GN_CHUNK - 1) / GN_CHUNK;
    return (size_t)N * G * chunks * 2 * sizeof(double);
}

int groupnorm_stats_launch(const float *x, float *stats, int N, int C, int S, int G, float eps, void *workspace,
                           hipStream_t s, const GnTable *tbl) {
    size_t cnt = (size_t)(C / G) * S;
    int chunks = (int)((cnt + GN_CHUNK - 1) / GN_CHUNK);
    GnTable t;   // (table == nullptr: statistics only)
    if (tbl && (N * G <= (int)RANGE_MAX_PARTS) && C / G <= 256) {
        t = *tbl;
        t.C = C;
        t.cpg = C / G;
        t.sqrt_ng = sqrtf((float)(C / G) * (float)S);
    }
    if (chunks <= GN_DIRECT_CHUNKS) {
        hipLaunchKernelGGL(gn_stats_direct_kernel, dim3(N * G), dim3(256), 0, s, x, stats, cnt, eps, t);
    } else {
        hipLaunchKernelGGL(gn_partial_kernel, dim3(chunks, N * G), dim3(256), 0, s, x, (double *)workspace, cnt, chunks);
        hipLaunchKernelGGL(gn_finalize_kernel, dim3(cdiv(N * G, 256)), dim3(256), 0, s, (const double *)workspace, stats,
                           N * G, chunks, (double)cnt, eps, t);
    }
    if (tbl && !t.table)   // (shape outside the fused form: the separate table launch)
        hipLaunchKernelGGL(gn_affine_table_kernel, dim3(cdiv((long)N * C, 256)), dim3(256), 0, s, stats, tbl->gamma, tbl->beta, tbl->w2, tbl->b2,
                           tbl->table, tbl->range, N, C, C / G, sqrtf((float)(C / G) * (float)S));
    return check_launch("groupnorm_stats");
}

int groupnorm_stats_from_tiles(const float *part, const float *bias, float *stats, int N, int C, int S, int G, float eps, int tiles_per_frame,
                               int waves, hipStream_t s, const GnTable *tbl) {
    GnTable t;
    if (tbl && (N * G <= (int)RANGE_MAX_PARTS) && C / G <= 256) {
        t = *tbl;
        t.C = C;
        t.cpg = C / G;
        t.sqrt_ng = sqrtf((float)(C / G) * (float)S);
    }
    hipLaunchKernelGGL(gn_tile_finalize_kernel, dim3(N * G), dim3(256), 0, s, part, bias, stats, tiles_per_frame, waves, C, C / G,
                       (double)(C / G) * (double)S, (double)S / ((double)tiles_per_frame * waves), eps, t);
    if (tbl && !t.table)
        hipLaunchKernelGGL(gn_affine_table_kernel, dim3(cdiv((long)N * C, 256)), dim3(256), 0, s, stats, tbl->gamma, tbl->beta, tbl->w2, tbl->b2,
                           tbl->table, tbl->range, N, C, C / G, sqrtf((float)(C / G) * (float)S));
    return check_launch("groupnorm_stats(conv tiles)");
}
}  // namespace mphip

extern "C" size_t mphip_groupnorm_workspace_bytes(int N, int C, int S, int G) { return groupnorm_ws_bytes(N, C, S, G); }

extern "C" int mphip_groupnorm_stats(const float *x, float *stats, int N, int C, int S, int G, float eps,
                                     void *workspace, size_t workspace_bytes, void *stream) {
    MPHIP_REQUIRE(x && stats, "groupnorm_stats: null pointer");
    MPHIP_REQUIRE(N > 0 && C > 0 && S > 0 && G > 0 && C % G == 0, "groupnorm_stats: bad dims (C=%d G=%d)", C, G);
    size_t need = groupnorm_ws_bytes(N, C, S, G);
    if (!workspace || workspace_bytes < need) {
        set_error("groupnorm_stats: workspace %zu bytes < required %zu", workspace_bytes, need);
        return MPHIP_EWORKSPACE;
    }
    return groupnorm_stats_launch(x, stats, N, C, S, G, eps, workspace, (hipStream_t)stream);
}

// workgroups for `total` threads of 256; capped when a range descriptor is filled (see gn_apply_kernel)
constexpr int RANGE_GRID_CAP = (int)RANGE_MAX_PARTS;
static inline int range_grid(size_t total, const float *range) {
    const int full = cdiv(total, 256);
    return (range && full > RANGE_GRID_CAP) ? RANGE_GRID_CAP : full;
}

extern "C" int mphip_groupnorm_apply(const float *x, const float *stats, const float *gamma, const float *beta,
                                     const float *w2, const float *b2, const float *residual, float *y, float *out_range, int N,
                                     int C, int D, int H, int W, int G, int relu, int tanh_, int pool2, void *stream) {
    MPHIP_REQUIRE(x && stats && gamma && beta && y, "groupnorm_apply: null pointer");
    MPHIP_REQUIRE(N > 0 && C > 0 && D > 0 && H > 0 && W > 0 && G > 0 && C % G == 0, "groupnorm_apply: bad dims");
    MPHIP_REQUIRE((w2 == nullptr) == (b2 == nullptr), "groupnorm_apply: w2/b2 must both be set or both NULL");
    GnParams p{x, stats, gamma, beta, w2, b2, residual, y, C, C / G, relu, tanh_, out_range};
    hipStream_t s = (hipStream_t)stream;
    const int S = D * H * W;
    if (pool2) {
        MPHIP_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, "groupnorm_apply: pool2 needs even D,H,W");
        size_t total = (size_t)N * C * (D / 2) * (H / 2) * (W / 2);
        if (W % 4 == 0 && (((uintptr_t)p.x | (uintptr_t)p.y | (uintptr_t)p.residual) & 15) == 0) {
            total /= 2;
            hipLaunchKernelGGL(gn_apply_pool_kernel<2>, dim3(range_grid(total, out_range)), dim3(256), 0, s, p, D, H, W, total);
        } else {
            hipLaunchKernelGGL(gn_apply_pool_kernel<1>, dim3(range_grid(total, out_range)), dim3(256), 0, s, p, D, H, W, total);
        }
    } else if (S % 4 == 0) {
        size_t total = (size_t)N * C * S / 4;
        hipLaunchKernelGGL(gn_apply_kernel<4>, dim3(range_grid(total, out_range)), dim3(256), 0, s, p, S, total);
    } else {
        size_t total = (size_t)N * C * S;
        hipLaunchKernelGGL(gn_apply_kernel<1>, dim3(range_grid(total, out_range)), dim3(256), 0, s, p, S, total);
    }
    return check_launch("groupnorm_apply");
}

// table[n][c] = (scale, shift) with GroupNorm(+AdaptiveGroupNorm's second affine) folded to y = x*scale + shift.
// range: a rigorous bound of max|y| that needs no pass over x — the normalised value obeys |x - mean| * rstd <= sqrt(Ng)
// (Ng = elements per group: sum (x-mean)^2 = Ng*var, rstd = 1/sqrt(var+eps)), so |y[c]| <= sqrt(Ng)*|gamma*w2| + |beta*w2 + b2|.
// Loose by ~100x for real activations, which costs nothing: the f16x3 split still resolves 2^-24 of the bound.
__global__ void gn_affine_table_kernel(const float *__restrict__ stats, const float *__restrict__ gamma,
                                       const float *__restrict__ beta, const float *__restrict__ w2,
                                       const float *__restrict__ b2, float *__restrict__ table, float *__restrict__ range, int N,
                                       int C, int cpg, float sqrt_ng) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned mbits = 0;
    if (i < N * C) {
        const int c = i % C, n = i / C;
        const int grp = n * (C / cpg) + c / cpg;
        const float mean = stats[grp * 2], rstd = stats[grp * 2 + 1];
        float scale = rstd * gamma[c];
        float shift = beta[c] - mean * scale;
        float amp = gamma[c], off = beta[c];
        if (w2) {
            scale = scale * w2[c];
            shift = shift * w2[c] + b2[c];
            amp = amp * w2[c];
            off = off * w2[c] + b2[c];
        }
        table[i * 2] = scale;
        table[i * 2 + 1] = shift;
        mbits = range_bits((sqrt_ng * fabsf(amp) + fabsf(off)) * 1.0001f);
    }
    if (range) range_note_block(mbits, range, blockIdx.x, gridDim.x);
}

extern "C" int mphip_groupnorm_affine_table(const float *stats, const float *gamma, const float *beta, const float *w2,
                                            const float *b2, float *table, float *out_range, int N, int C, int S, int G,
                                            void *stream) {
    MPHIP_REQUIRE(stats && gamma && beta && table, "groupnorm_affine_table: null pointer");
    MPHIP_REQUIRE(N > 0 && C > 0 && S > 0 && G > 0 && C % G == 0, "groupnorm_affine_table: bad dims");
    MPHIP_REQUIRE((w2 == nullptr) == (b2 == nullptr), "groupnorm_affine_table: w2/b2 must both be set or both NULL");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_affine_table_kernel, dim3(cdiv((long)N * C, 256)), dim3(256), 0, s, stats, gamma, beta, w2, b2, table,
                       out_range, N, C, C / G, sqrtf((float)(C / G) * (float)S));
    return check_launch("groupnorm_affine_table");
}

extern "C" int mphip_groupnorm_stats_split(const float *x, int x_splits, const float *x_bias, float *stats, int N, int C,
                                           int S, int G, float eps, void *stream) {
    MPHIP_REQUIRE(x && stats, "groupnorm_stats_split: null pointer");
    MPHIP_REQUIRE(N > 0 && C > 0 && S > 0 && G > 0 && C % G == 0 && x_splits >= 1, "groupnorm_stats_split: bad dims");
    MPHIP_REQUIRE((size_t)(C / G) * S <= (size_t)GN_DIRECT_CHUNKS * GN_CHUNK,
                  "groupnorm_stats_split: group span too large for the single-launch path (reduce first)");
    // one workgroup per (sample, group): long spans get 16 waves (a thread's elements are a serial chain of slab round trips)
    hipLaunchKernelGGL(gn_stats_split_kernel, dim3(N * G), dim3((size_t)(C / G) * S >= 2048 ? 1024 : 256), 0, (hipStream_t)stream, x, x_splits,
                       (size_t)N * C * S, x_bias, stats, C, C / G, S, eps);
    return check_launch("groupnorm_stats_split");
}

extern "C" int mphip_groupnorm_apply_split(const float *x, int x_splits, const float *x_bias, const float *stats,
                                           const float *gamma, const float *beta, const float *w2, const float *b2,
                                           const float *residual, int res_splits, const float *res_bias, float *y,
                                           float *out_range, int N, int C, int D, int H, int W, int G, int relu, int tanh_,
                                           int pool2, int uD, int uH, int uW, void *stream) {
    MPHIP_REQUIRE(x && stats && gamma && beta && y, "groupnorm_apply_split: null pointer");
    MPHIP_REQUIRE(N > 0 && C > 0 && D > 0 && H > 0 && W > 0 && G > 0 && C % G == 0, "groupnorm_apply_split: bad dims");
    MPHIP_REQUIRE((w2 == nullptr) == (b2 == nullptr), "groupnorm_apply_split: w2/b2 must both be set or both NULL");
    MPHIP_REQUIRE(x_splits >= 1 && res_splits >= 1 && uD >= 1 && uH >= 1 && uW >= 1, "groupnorm_apply_split: bad split/up");
    MPHIP_REQUIRE(!(pool2 && (uD * uH * uW != 1)), "groupnorm_apply_split: pool2 and upsampling are exclusive");
    MPHIP_REQUIRE(!pool2 || (D % 2 == 0 && H % 2 == 0 && W % 2 == 0), "groupnorm_apply_split: pool2 needs even D,H,W");
    GnSplitParams q{{x, stats, gamma, beta, w2, b2, residual, y, C, C / G, relu, tanh_, out_range}, x_splits, res_splits,
                    (size_t)N * C * D * H * W, x_bias, res_bias, D, H, W, pool2, uD, uH, uW};
    size_t total = pool2 ? (size_t)N * C * (D / 2) * (H / 2) * (W / 2) : (size_t)N * C * D * uD * H * uH * W * uW;
    hipLaunchKernelGGL(gn_apply_split_kernel, dim3(range_grid(total, out_range)), dim3(256), 0, (hipStream_t)stream, q, total);
    return check_launch("groupnorm_apply_split");
}

extern "C" int mphip_groupnorm_small_fused(const float *x, int x_splits, const float *x_bias, const float *gamma,
                                          const float *beta, const float *w2, const float *b2, const float *residual,
                                          int res_splits, const float *res_bias, float *y, float *stats_out, int N, int C,
                                          int D, int H, int W, int G, float eps, int relu, int tanh_, int uD, int uH, int uW,
                                          void *stream) {
    MPHIP_REQUIRE(x && gamma && beta && y, "groupnorm_small_fused: null pointer");
    MPHIP_REQUIRE(N > 0 && C > 0 && D > 0 && H > 0 && W > 0 && G > 0 && C % G == 0, "groupnorm_small_fused: bad dims");
    MPHIP_REQUIRE((w2 == nullptr) == (b2 == nullptr), "groupnorm_small_fused: w2/b2 must both be set or both NULL");
    MPHIP_REQUIRE(x_splits >= 1 && res_splits >= 1 && uD >= 1 && uH >= 1 && uW >= 1, "groupnorm_small_fused: bad split/up");
    MPHIP_REQUIRE((size_t)(C / G) * D * H * W <= (size_t)GN_FUSED_MAX,
                  "groupnorm_small_fused: (sample, group) span %zu exceeds %d floats", (size_t)(C / G) * D * H * W, GN_FUSED_MAX);
    GnSplitParams q{{x, nullptr, gamma, beta, w2, b2, residual, y, C, C / G, relu, tanh_}, x_splits, res_splits,
                    (size_t)N * C * D * H * W, x_bias, res_bias, D, H, W, 0, uD, uH, uW};
    const int threads = (N * G < 64 && (size_t)(C / G) * D * H * W >= 4096) ? 1024 : 256;
    const size_t lds = 272 + (size_t)(C / G) * D * H * W * sizeof(float);
    hipLaunchKernelGGL(gn_small_fused_kernel, dim3(N * G), dim3(threads), lds, (hipStream_t)stream, q, eps, stats_out);
    return check_launch("groupnorm_small_fused");
}

extern "C" int mphip_avgpool2(const float *x, float *y, int NC, int D, int H, int W, void *stream) {
    MPHIP_REQUIRE(x && y, "avgpool2: null pointer");
    MPHIP_REQUIRE(NC > 0 && D > 0 && H > 0 && W > 0 && D % 2 == 0 && H % 2 == 0 && W % 2 == 0,
                  "avgpool2: dims must be positive and even");
    size_t total = (size_t)NC * (D / 2) * (H / 2) * (W / 2);
    hipLaunchKernelGGL(avgpool2_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, D, H, W, total);
    return check_launch("avgpool2");
}

extern "C" int mphip_upsample_trilinear2(const float *x, float *y, int NC, int D, int H, int W, void *stream) {
    MPHIP_REQUIRE(x && y, "upsample_trilinear2: null pointer");
    MPHIP_REQUIRE(NC > 0 && D > 0 && H > 0 && W > 0, "upsample_trilinear2: bad dims");
    size_t total = (size_t)NC * D * H * W * 8;
    if (W % 2 == 0) {
        // align_corners=True scale, one fp32 division per axis exactly as ATen computes it
        const float sD = 2 * D > 1 ? (float)(D - 1) / (float)(2 * D - 1) : 0.0f;
        const float sH = 2 * H > 1 ? (float)(H - 1) / (float)(2 * H - 1) : 0.0f;
        const float sW = 2 * W > 1 ? (float)(W - 1) / (float)(2 * W - 1) : 0.0f;
        // (an LDS-staged variant of this kernel — 7 coalesced staging loads per thread instead of 36 scalar ones — measured
        //  +-0 on the 201 MB upsample, r03: the kernel is bound by its write stream, not by load issue)
        hipLaunchKernelGGL(upsample_trilinear2_kernel, dim3(cdiv(total / 16, 256)), dim3(256), 0, (hipStream_t)stream, x, y, D,
                           H, W, sD, sH, sW, total / 16);
    } else {
        hipLaunchKernelGGL(upsample_trilinear2_scalar_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y,
                           D, H, W, total);
    }
    return check_launch("upsample_trilinear2");
}

// The x2 upsample restricted to what a demand-driven conv will read (see mphip_conv3d_fwd_roi): x [N*C, D,H,W] -> the parts of
// y [N*C, 2D,2H,2W] inside the 1-voxel halo of every (tD,tH,tW) conv tile a sample box touches; the rest of y is left untouched.
extern "C" int mphip_upsample_trilinear2_roi(const float *x, float *y, const int *roi, int roi_frames, int N, int C, int D, int H, int W,
                                             int tD, int tH, int tW, void *stream) {
    MPHIP_REQUIRE(x && y && roi, "upsample_trilinear2_roi: null pointer");
    MPHIP_REQUIRE(N > 0 && C > 0 && D > 0 && H > 0 && W > 0 && tD > 0 && tH > 0 && tW > 0 && W % 2 == 0, "upsample_trilinear2_roi: bad dims");
    MPHIP_REQUIRE(roi_frames >= 0 && (roi_frames == 0 || N == 1), "upsample_trilinear2_roi: roi_frames > 0 needs N == 1");
    const size_t total = (size_t)N * C * D * H * W * 8;
    const float sD = 2 * D > 1 ? (float)(D - 1) / (float)(2 * D - 1) : 0.0f;
    const float sH = 2 * H > 1 ? (float)(H - 1) / (float)(2 * H - 1) : 0.0f;
    const float sW = 2 * W > 1 ? (float)(W - 1) / (float)(2 * W - 1) : 0.0f;
    (void)total;
    hipLaunchKernelGGL(upsample_trilinear2_roi_kernel, dim3((unsigned)((size_t)N * C)), dim3(256), 0, (hipStream_t)stream, x, y, D, H, W, sD, sH,
                       sW, roi, roi_frames, C, tD, tH, tW);
    return check_launch("upsample_trilinear2_roi");
}

extern "C" int mphip_upsample_trilinear(const float *x, float *y, int NC, int D, int H, int W, int sD, int sH, int sW,
                                        void *stream) {
    MPHIP_REQUIRE(x && y, "upsample_trilinear: null pointer");
    MPHIP_REQUIRE(NC > 0 && D > 0 && H > 0 && W > 0 && sD > 0 && sH > 0 && sW > 0, "upsample_trilinear: bad dims");
    size_t total = (size_t)NC * D * H * W * sD * sH * sW;
    hipLaunchKernelGGL(upsample_trilinear_scaled_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, D, H, W,
                       sD, sH, sW, total);
    return check_launch("upsample_trilinear");
}

extern "C" int mphip_upsample_nearest(const float *x, float *y, int NC, int D, int H, int W, int sD, int sH, int sW,
                                      void *stream) {
    MPHIP_REQUIRE(x && y, "upsample_nearest: null pointer");
    MPHIP_REQUIRE(NC > 0 && D > 0 && H > 0 && W > 0 && sD > 0 && sH > 0 && sW > 0, "upsample_nearest: bad dims");
    size_t total = (size_t)NC * D * H * W * sD * sH * sW;
    hipLaunchKernelGGL(upsample_nearest_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, D, H, W,
                       sD, sH, sW, total);
    return check_launch("upsample_nearest");
}

extern "C" int mphip_add_matmul(const float *a, const float *a2, const float *m, const float *bias, float *out, int B,
                                int K, int N, int trans, void *stream) {
    MPHIP_REQUIRE(a && m && out, "add_matmul: null pointer");
    MPHIP_REQUIRE(B > 0 && K > 0 && N > 0, "add_matmul: bad dims");
    dim3 grid(cdiv(N, 4));
    if (trans) {
        hipLaunchKernelGGL(add_matmul_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a, a2, m, bias, out, B, K, N);
    } else if (K <= MM_KMAX) {
        for (int b0 = 0; b0 < B; b0 += MM_BMAX)
            hipLaunchKernelGGL(add_matmul_kn_kernel, dim3(cdiv(N, MM_COLS)), dim3(MM_WAVES * 64), 0, (hipStream_t)stream, a, a2, m,
                               bias, out, B, K, N, b0);
    } else {
        hipLaunchKernelGGL(add_matmul_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a, a2, m, bias, out, B, K, N);
    }
    return check_launch("add_matmul");
}
