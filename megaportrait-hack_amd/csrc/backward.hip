// Backward kernels of the G3d blocks (scope row f2, SURVEY.md §8f: training through the HIP path).
// First, correctness-oriented versions: every gradient the reference's autograd produces for
// nn.Conv3d / nn.GroupNorm+ReLU(+residual) / nn.AvgPool3d / nn.Upsample(trilinear, align_corners=True)
// (model.py:500-528, 571-597) has a HIP kernel here; bwd-data of the convs reuses the forward conv kernels
// on flipped/transposed weights (host side), so it already runs on the f16x3 path.
#include "mphip_common.h"
#include "mphip_conv.h"
#include "mphip_resample.h"

namespace mphip {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------------
// conv3d 3x3x3 / 1x1x1 backward-weight:  dW[co][ci][tap] = sum_{n,v} dY[n][co][v] * X[n][ci][v + tap]
// GEMM per tap: M = co, N = ci, K = voxels, on v_mfma_f32_32x32x2_f32 (exact fp32).
// Workgroup = 3 waves, one (96-co tile, 32-ci tile, kd plane) and a slice of the voxel range; wave w owns the
// kh = w row of the plane (3 taps x 3 co tiles = 9 accumulators).  Per 1x8x8 voxel tile the dY tile
// [96][64] and the X halo tile [32][3][10][10] (one kd plane deep: only d+kd-1 is needed) are staged in
// LDS with odd pitches (lanes walk channels: conflict-free).  Partial sums go to slab[blockIdx.z] and are
// reduced in order by conv_bwd_weight_reduce_kernel (deterministic).
constexpr int BW_VT = 64;          // voxels per staged tile (1 x 8 x 8)
constexpr int BW_DYP = BW_VT + 1;  // dY pitch
constexpr int BW_XV = 10 * 10;     // halo voxels of one depth slice (8+2)^2
constexpr int BW_XP = BW_XV + 1;   // X pitch

template <int KS>
__global__ void __launch_bounds__(192)
conv_bwd_weight_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ slabs, int N, int Ci,
                       int Co, int D, int H, int W, int tiles_per_split, unsigned x_bytes) {
    __shared__ float dys[96 * BW_DYP];
    __shared__ float xs[32 * BW_XP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kk = lane >> 5;
    const int HW = H * W, DHW = D * HW;
    const int ci_tiles = (Ci + 31) / 32;
    const int ci0 = (blockIdx.x % ci_tiles) * 32;
    const int co0 = (blockIdx.x / ci_tiles) * 96;
    const int kd = KS == 3 ? blockIdx.y : 1;            // tap plane (depth offset kd-1)
    const int kh = KS == 3 ? wave : 1;                   // this wave's row of the plane
    const int tiles_h = (H + 7) / 8, tiles_w = (W + 7) / 8;
    const long ntiles = (long)N * D * tiles_h * tiles_w;
    const long t_begin = (long)blockIdx.z * tiles_per_split;
    const long t_end = min(ntiles, t_begin + tiles_per_split);
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)x_bytes, 0x00020000);

    constexpr int NTAP = KS == 3 ? 3 : 1;  // taps per wave (kw = 0..2)
    constexpr int MT = KS == 3 ? 3 : 1;    // co tiles per wave: k=3 -> all three, k=1 -> tile `wave`
    const int m0 = KS == 3 ? 0 : wave;
    f32x16 acc[NTAP][MT];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][m][r] = 0.0f;

    for (long tile = t_begin; tile < t_end; ++tile) {
        long r = tile;
        const int tw = (int)(r % tiles_w); r /= tiles_w;
        const int th = (int)(r % tiles_h); r /= tiles_h;
        const int d = (int)(r % D);
        const int n = (int)(r / D);
        const int h0 = th * 8, w0 = tw * 8;
        __syncthreads();  // previous tile fully consumed
        // dY tile [96 co][64 vox]
        for (int e = tid; e < 96 * BW_VT; e += 192) {
            const int c = e / BW_VT, v = e % BW_VT;
            const int co = co0 + c;
            float val = 0.0f;
            const int gh = h0 + v / 8, gw = w0 + v % 8;  // ragged edge tiles contribute zeros
            if (co < Co && gh < H && gw < W) val = dy[((size_t)n * Co + co) * DHW + (size_t)d * HW + gh * W + gw];
            dys[c * BW_DYP + v] = val;
        }
        // X slice d+kd-1, halo 10x10, 32 channels (zero outside the volume / beyond Ci)
        const int xd = d + kd - 1;
        for (int e = tid; e < 32 * BW_XV; e += 192) {
            const int c = e / BW_XV, q = e % BW_XV;
            const int gh = h0 - 1 + q / 10, gw = w0 - 1 + q % 10;
            const int ci = ci0 + c;
            unsigned off = OOB;
            if (ci < Ci && (unsigned)xd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W)
                off = (unsigned)((((long)n * Ci + ci) * DHW + (long)xd * HW + gh * W + gw) * 4);
            xs[c * BW_XP + q] = buf_load_f(rsrc, off, 0);
        }
        __syncthreads();
        {
#pragma unroll 4
            for (int ks = 0; ks < BW_VT / 2; ++ks) {
                const int v = 2 * ks + kk;                 // this lane's voxel (k index of the MFMA)
                const int vh = v / 8, vw = v % 8;
                float a[MT], b[NTAP];
#pragma unroll
                for (int m = 0; m < MT; ++m) a[m] = dys[((m0 + m) * 32 + j) * BW_DYP + v];  // A[i=co][k=vox]
#pragma unroll
                for (int t = 0; t < NTAP; ++t) {
                    const int kw = KS == 3 ? t : 1;
                    b[t] = xs[j * BW_XP + (vh + kh) * 10 + vw + kw];                       // B[k=vox][j=ci]
                }
#pragma unroll
                for (int t = 0; t < NTAP; ++t)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[t], acc[t][m], 0, 0, 0);
            }
        }
    }
    // slab layout = OIDHW gradient [Co][Ci][taps]; C/D: col = lane&31 = ci, row = co
    constexpr int TAPS = KS * KS * KS;
    float *slab = slabs + (size_t)blockIdx.z * Co * Ci * TAPS;
    const int ci = ci0 + j;
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
        const int tap = KS == 3 ? (kd * 3 + kh) * 3 + t : 0;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int co = co0 + (m0 + m) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kk;
                if (co < Co && ci < Ci) slab[((size_t)co * Ci + ci) * TAPS + tap] = acc[t][m][reg];
            }
    }
}

__global__ void __launch_bounds__(256)
slab_reduce_kernel(const float *__restrict__ slabs, float *__restrict__ out, size_t n, int splits) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = sum_slabs(slabs, splits, n, i);
}

// ---------------------------------------------------------------------------------------------------
// GroupNorm (+residual, +ReLU) backward.  Forward: xh = (x-mean)*rstd; z = xh*gamma+beta; u = z (+res); y = relu(u).
// Pass 1 (one workgroup per (n,c) plane): du = dy * (y > 0) [no mask when relu == 0];
//         s1[n,c] = sum du, s2[n,c] = sum du*xh.
// (host, tiny [N,C] tensors: dbeta = sum_n s1, dgamma = sum_n s2, A[n,g] = sum_c gamma_c*s1/cnt, B[n,g] = sum_c gamma_c*s2/cnt)
// Pass 2: dx = rstd * (gamma_c*du - A - xh*B);  dres = du.
// gradient through the activation that follows the norm: act 0 = none, 1 = ReLU, 2 = tanh(ReLU(.)) (FlowField's head,
// model.py:462-465); y is the forward output (y > 0 <=> pre-activation > 0; d tanh = 1 - y^2)
__device__ __forceinline__ float act_grad(float dy, float y, int act) {
    if (act == 0) return dy;
    if (!(y > 0.0f)) return 0.0f;
    return act == 2 ? dy * (1.0f - y * y) : dy;
}

constexpr int GNB_CHUNK = 8192;  // floats per workgroup of the reduce pass
__global__ void __launch_bounds__(256)
gn_bwd_reduce_kernel(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ dy,
                     const float *__restrict__ stats, float *__restrict__ partial, int C, int cpg, int S, int relu, int chunks) {
    const int plane = blockIdx.x / chunks, chunk = blockIdx.x % chunks;  // plane = n*C + c
    const int c = plane % C, n = plane / C;
    const int grp = n * (C / cpg) + c / cpg;
    const float mean = stats[grp * 2], rstd = stats[grp * 2 + 1];
    const size_t base = (size_t)plane * S;
    const int begin = chunk * GNB_CHUNK, end = min(S, begin + GNB_CHUNK);
    float s1 = 0.0f, s2 = 0.0f;
    if ((S & 3) == 0) {
        for (int i = begin + threadIdx.x * 4; i < end; i += 1024) {
            const float4 g = *reinterpret_cast<const float4 *>(dy + base + i);
            const float4 xv = *reinterpret_cast<const float4 *>(x + base + i);
            float du[4] = {g.x, g.y, g.z, g.w};
            if (relu) {
                const float4 yv = *reinterpret_cast<const float4 *>(y + base + i);
                du[0] = act_grad(du[0], yv.x, relu);
                du[1] = act_grad(du[1], yv.y, relu);
                du[2] = act_grad(du[2], yv.z, relu);
                du[3] = act_grad(du[3], yv.w, relu);
            }
            const float xh[4] = {(xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd};
            s1 += (du[0] + du[1]) + (du[2] + du[3]);
            s2 += (du[0] * xh[0] + du[1] * xh[1]) + (du[2] * xh[2] + du[3] * xh[3]);
        }
    } else {
        for (int i = begin + threadIdx.x; i < end; i += 256) {
            const float du = relu ? act_grad(dy[base + i], y[base + i], relu) : dy[base + i];
            s1 += du;
            s2 += du * ((x[base + i] - mean) * rstd);
        }
    }
    double d1 = (double)s1, d2 = (double)s2;  // <= 32 fp32 terms per thread, combined in double
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) {
        d1 += __shfl_xor(d1, sft, 64);
        d2 += __shfl_xor(d2, sft, 64);
    }
    __shared__ double red[8];
    if ((threadIdx.x & 63) == 0) {
        red[(threadIdx.x >> 6) * 2] = d1;
        red[(threadIdx.x >> 6) * 2 + 1] = d2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[(size_t)blockIdx.x * 2] = (float)((red[0] + red[2]) + (red[4] + red[6]));
        partial[(size_t)blockIdx.x * 2 + 1] = (float)((red[1] + red[3]) + (red[5] + red[7]));
    }
}

// folds the partial sums (one workgroup): s12[n][c] = sum over chunks; with S1 = sum_n s1, S2 = sum_n s2 and the
// AdaptiveGroupNorm second affine w2 (model.py:314-316; 1 if absent): dgamma = w2*S2, dbeta = w2*S1,
// dw2 = gamma*S2 + beta*S1, db2 = S1;  ab[n][g] = (sum_{c in g} gamma_c*w2_c*s1, sum ... *s2) / (cpg*S)
__global__ void __launch_bounds__(1024)
gn_bwd_fold_kernel(const float *__restrict__ partial, const float *__restrict__ gamma, const float *__restrict__ beta,
                   const float *__restrict__ w2, float *__restrict__ s12, float *__restrict__ dgamma, float *__restrict__ dbeta,
                   float *__restrict__ dw2, float *__restrict__ db2, float *__restrict__ ab, int N, int C, int cpg, int S,
                   int chunks) {
    for (int p = threadIdx.x; p < N * C; p += 1024) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < chunks; ++k) {
            a += (double)partial[((size_t)p * chunks + k) * 2];
            b += (double)partial[((size_t)p * chunks + k) * 2 + 1];
        }
        s12[p * 2] = (float)a;
        s12[p * 2 + 1] = (float)b;
    }
    __syncthreads();  // one workgroup: its own global writes are visible after the barrier
    for (int c = threadIdx.x; c < C; c += 1024) {
        double a = 0.0, b = 0.0;
        for (int n = 0; n < N; ++n) {
            a += (double)s12[(n * C + c) * 2];
            b += (double)s12[(n * C + c) * 2 + 1];
        }
        const double w = w2 ? (double)w2[c] : 1.0;
        dbeta[c] = (float)(w * a);
        dgamma[c] = (float)(w * b);
        if (w2) {
            dw2[c] = (float)((double)gamma[c] * b + (double)beta[c] * a);
            db2[c] = (float)a;
        }
    }
    const int G = C / cpg;
    const double inv = 1.0 / ((double)cpg * (double)S);
    for (int q = threadIdx.x; q < N * G; q += 1024) {
        const int n = q / G, g = q % G;
        double a = 0.0, b = 0.0;
        for (int k = 0; k < cpg; ++k) {
            const int c = g * cpg + k;
            const double ge = (double)gamma[c] * (w2 ? (double)w2[c] : 1.0);
            a += ge * (double)s12[(n * C + c) * 2];
            b += ge * (double)s12[(n * C + c) * 2 + 1];
        }
        ab[q * 2] = (float)(a * inv);
        ab[q * 2 + 1] = (float)(b * inv);
    }
}

__global__ void __launch_bounds__(256)
gn_bwd_apply_kernel(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ dy,
                    const float *__restrict__ stats, const float *__restrict__ gamma, const float *__restrict__ w2,
                    const float *__restrict__ ab, float *__restrict__ dx, float *__restrict__ dres, int C, int cpg, int S,
                    int relu, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t plane = i / S;
    const int c = (int)(plane % C), n = (int)(plane / C);
    const int grp = n * (C / cpg) + c / cpg;
    const float mean = stats[grp * 2], rstd = stats[grp * 2 + 1];
    const float du = relu ? act_grad(dy[i], y[i], relu) : dy[i];
    const float xh = (x[i] - mean) * rstd;
    const float ge = w2 ? gamma[c] * w2[c] : gamma[c];
    dx[i] = rstd * (ge * du - ab[grp * 2] - xh * ab[grp * 2 + 1]);
    if (dres) dres[i] = du;
}

// S % 4 == 0 (every tensor of the hot path): four consecutive elements of one (n,c) plane per thread — 16-byte loads / stores, one plane
// lookup per thread instead of a 64-bit division per element; the same expression per element as gn_bwd_apply_kernel (same bits).
// grid = (ceil(S / 1024), N * C)
__global__ void __launch_bounds__(256)
gn_bwd_apply4_kernel(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ dy,
                     const float *__restrict__ stats, const float *__restrict__ gamma, const float *__restrict__ w2,
                     const float *__restrict__ ab, float *__restrict__ dx, float *__restrict__ dres, int C, int cpg, int S, int relu) {
    const int plane = blockIdx.y, c = plane % C, n = plane / C;
    const int e = ((int)blockIdx.x * 256 + (int)threadIdx.x) * 4;
    if (e >= S) return;
    const int grp = n * (C / cpg) + c / cpg;
    const float mean = stats[grp * 2], rstd = stats[grp * 2 + 1];
    const float ge = w2 ? gamma[c] * w2[c] : gamma[c];
    const float a = ab[grp * 2], b = ab[grp * 2 + 1];
    const size_t i = (size_t)plane * S + e;
    const float4 g = *reinterpret_cast<const float4 *>(dy + i);
    const float4 xv = *reinterpret_cast<const float4 *>(x + i);
    float du[4] = {g.x, g.y, g.z, g.w};
    if (relu) {
        const float4 yv = *reinterpret_cast<const float4 *>(y + i);
        du[0] = act_grad(du[0], yv.x, relu);
        du[1] = act_grad(du[1], yv.y, relu);
        du[2] = act_grad(du[2], yv.z, relu);
        du[3] = act_grad(du[3], yv.w, relu);
    }
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float xh = (xs[k] - mean) * rstd;
        o[k] = rstd * (ge * du[k] - a - xh * b);
    }
    *reinterpret_cast<float4 *>(dx + i) = make_float4(o[0], o[1], o[2], o[3]);
    if (dres) *reinterpret_cast<float4 *>(dres + i) = make_float4(du[0], du[1], du[2], du[3]);
}

// ---- the fold folded into the apply (r05): mphip_groupnorm_bwd = reduce + ONE apply launch.  What gn_bwd_fold_kernel computed in a
// launch of its own (32 launches of ~5 us in a training step of the slice) is a few dozen numbers per consumer: every apply workgroup
// re-derives the (A, B) of its own (n, group) from the reduce pass's partial sums — cpg x chunks <= ~24 pairs, with the fold kernel's own
// operation order (float-rounded per-(n,c) sums of double accumulations, then a double sum over the group's channels): same bits — and
// one designated thread per channel writes dgamma / dbeta (/ dw2 / db2).
__device__ __forceinline__ void gn_fold_s12(const float *__restrict__ partial, int p, int chunks, float &s1, float &s2) {
    double a = 0.0, b = 0.0;
    for (int k = 0; k < chunks; ++k) {
        a += (double)partial[((size_t)p * chunks + k) * 2];
        b += (double)partial[((size_t)p * chunks + k) * 2 + 1];
    }
    s1 = (float)a;
    s2 = (float)b;
}
__device__ __forceinline__ void gn_fold_ab(const float *__restrict__ partial, const float *__restrict__ gamma, const float *__restrict__ w2,
                                           int n, int g, int C, int cpg, int S, int chunks, float &A, float &B) {
    const double inv = 1.0 / ((double)cpg * (double)S);
    double a = 0.0, b = 0.0;
    for (int k = 0; k < cpg; ++k) {
        const int c = g * cpg + k;
        float s1, s2;
        gn_fold_s12(partial, n * C + c, chunks, s1, s2);
        const double ge = (double)gamma[c] * (w2 ? (double)w2[c] : 1.0);
        a += ge * (double)s1;
        b += ge * (double)s2;
    }
    A = (float)(a * inv);
    B = (float)(b * inv);
}
__device__ __forceinline__ void gn_fold_channel(const float *__restrict__ partial, const float *__restrict__ gamma, const float *__restrict__ beta,
                                                const float *__restrict__ w2, int c, int N, int C, int chunks, float *__restrict__ dgamma,
                                                float *__restrict__ dbeta, float *__restrict__ dw2, float *__restrict__ db2) {
    double a = 0.0, b = 0.0;
    for (int n = 0; n < N; ++n) {
        float s1, s2;
        gn_fold_s12(partial, n * C + c, chunks, s1, s2);
        a += (double)s1;
        b += (double)s2;
    }
    const double w = w2 ? (double)w2[c] : 1.0;
    dbeta[c] = (float)(w * a);
    dgamma[c] = (float)(w * b);
    if (w2) {
        dw2[c] = (float)((double)gamma[c] * b + (double)beta[c] * a);
        db2[c] = (float)a;
    }
}

// vector form (S % 4 == 0, S >= 512): grid = (ceil(S / 1024), N * C); (n, group) is workgroup-uniform
__global__ void __launch_bounds__(256)
gn_bwd_apply4_fold_kernel(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ dy,
                          const float *__restrict__ stats, const float *__restrict__ gamma, const float *__restrict__ beta,
                          const float *__restrict__ w2, const float *__restrict__ partial, float *__restrict__ dx, float *__restrict__ dres,
                          float *__restrict__ dgamma, float *__restrict__ dbeta, float *__restrict__ dw2, float *__restrict__ db2, int N, int C,
                          int cpg, int S, int relu, int chunks) {
    const int plane = blockIdx.y, c = plane % C, n = plane / C;
    if (blockIdx.x == 0 && n == 0 && threadIdx.x == 0) gn_fold_channel(partial, gamma, beta, w2, c, N, C, chunks, dgamma, dbeta, dw2, db2);
    const int e = ((int)blockIdx.x * 256 + (int)threadIdx.x) * 4;
    if (e >= S) return;
    const int grp = n * (C / cpg) + c / cpg;
    float a, b;
    gn_fold_ab(partial, gamma, w2, n, c / cpg, C, cpg, S, chunks, a, b);
    const float mean = stats[grp * 2], rstd = stats[grp * 2 + 1];
    const float ge = w2 ? gamma[c] * w2[c] : gamma[c];
    const size_t i = (size_t)plane * S + e;
    const float4 g = *reinterpret_cast<const float4 *>(dy + i);
    const float4 xv = *reinterpret_cast<const float4 *>(x + i);
    float du[4] = {g.x, g.y, g.z, g.w};
    if (relu) {
        const float4 yv = *reinterpret_cast<const float4 *>(y + i);
        du[0] = act_grad(du[0], yv.x, relu);
        du[1] = act_grad(du[1], yv.y, relu);
        du[2] = act_grad(du[2], yv.z, relu);
        du[3] = act_grad(du[3], yv.w, relu);
    }
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float xh = (xs[k] - mean) * rstd;
        o[k] = rstd * (ge * du[k] - a - xh * b);
    }
    *reinterpret_cast<float4 *>(dx + i) = make_float4(o[0], o[1], o[2], o[3]);
    if (dres) *reinterpret_cast<float4 *>(dres + i) = make_float4(du[0], du[1], du[2], du[3]);
}

// flat form (any S): one thread per element; the first C threads of the grid also write their channel's parameter gradients
__global__ void __launch_bounds__(256)
gn_bwd_apply_fold_kernel(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ dy,
                         const float *__restrict__ stats, const float *__restrict__ gamma, const float *__restrict__ beta,
                         const float *__restrict__ w2, const float *__restrict__ partial, float *__restrict__ dx, float *__restrict__ dres,
                         float *__restrict__ dgamma, float *__restrict__ dbeta, float *__restrict__ dw2, float *__restrict__ db2, int N, int C,
                         int cpg, int S, int relu, int chunks, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)C) gn_fold_channel(partial, gamma, beta, w2, (int)i, N, C, chunks, dgamma, dbeta, dw2, db2);
    if (i >= total) return;
    const size_t plane = i / S;
    const int c = (int)(plane % C), n = (int)(plane / C);
    const int grp = n * (C / cpg) + c / cpg;
    float a, b;
    gn_fold_ab(partial, gamma, w2, n, c / cpg, C, cpg, S, chunks, a, b);
    const float mean = stats[grp * 2], rstd = stats[grp * 2 + 1];
    const float du = relu ? act_grad(dy[i], y[i], relu) : dy[i];
    const float xh = (x[i] - mean) * rstd;
    const float ge = w2 ? gamma[c] * w2[c] : gamma[c];
    dx[i] = rstd * (ge * du - a - xh * b);
    if (dres) dres[i] = du;
}

// AvgPool3d(2,2) backward: dx[2d+a][2h+b][2w+c] = dout[d][h][w] / 8
__global__ void __launch_bounds__(256)
avgpool2_bwd_kernel(const float *__restrict__ dout, float *__restrict__ dx, int D, int H, int W, size_t total) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // over dx elements
    if (t >= total) return;
    const int w = (int)(t % W);
    size_t r = t / W;
    const int h = (int)(r % H);
    r /= H;
    const int d = (int)(r % D);
    const size_t plane = r / D;
    dx[t] = dout[((plane * (D / 2) + d / 2) * (H / 2) + h / 2) * (W / 2) + w / 2] / 8.0f;
}

// nn.Upsample(x2, trilinear, align_corners=True) backward (adjoint of upsample_trilinear2): each input voxel
// gathers, per axis, the <= 5 outputs whose source interval touches it, with the forward's own weights
// (deterministic, no atomics).  dx[i] = sum_o w(o,i) * dout[o].
// per-axis adjoint taps of input index i: the outputs o with i0(o) == i or i1(o) == i lie in
// ((i-1)/scale, (i+1)/scale): <= 5 consecutive o (scale ~ 1/2; 6 slots: one spare for rounding of the bound);
// w[k] = the forward's weight of (lo+k -> i), 0 if none
struct AdjTaps {
    int lo;
    float w[6];
};
__device__ __forceinline__ AdjTaps adj_taps(int i, int in, float scale, int out) {
    AdjTaps t;
    t.lo = scale > 0.0f ? max(0, (int)floorf(((float)i - 1.0f) / scale)) : 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int o = t.lo + k;
        float w = 0.0f;
        if (o < out) {
            const float src = scale * (float)o;
            const int i0 = min((int)src, in - 1);
            const int i1 = i0 + (i0 < in - 1 ? 1 : 0);
            const float l1 = src - (float)i0, l0 = 1.0f - l1;
            if (i0 == i) w += l0;
            if (i1 == i) w += l1;
        }
        t.w[k] = w;
    }
    return t;
}
__global__ void __launch_bounds__(256)
upsample_trilinear2_bwd_kernel(const float *__restrict__ dout, float *__restrict__ dx, int D, int H, int W, float sD,
                               float sH, float sW, size_t total) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // over dx (input-sized) elements
    if (t >= total) return;
    const int w = (int)(t % W);
    size_t r = t / W;
    const int h = (int)(r % H);
    r /= H;
    const int d = (int)(r % D);
    const size_t plane = r / D;
    const int oD = 2 * D, oH = 2 * H, oW = 2 * W;
    const AdjTaps td = adj_taps(d, D, sD, oD), th = adj_taps(h, H, sH, oH), tw = adj_taps(w, W, sW, oW);
    const float *p = dout + plane * (size_t)oD * oH * oW;
    float acc = 0.0f;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        if (td.w[a] == 0.0f) continue;
        float pl = 0.0f;
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            if (th.w[b] == 0.0f) continue;
            const float *row = p + ((size_t)(td.lo + a) * oH + th.lo + b) * oW + tw.lo;
            float rs = 0.0f;
#pragma unroll
            for (int c = 0; c < 6; ++c)
                if (tw.w[c] != 0.0f) rs += tw.w[c] * row[c];
            pl += th.w[b] * rs;
        }
        acc += td.w[a] * pl;
    }
    dx[t] = acc;
}

// One axis of the same adjoint (the trilinear resize is separable): gin[outer][i][inner] = sum_o w(o -> i) * gout[outer][o][inner],
// out_len = 2 * in_len.  Three of these (D, then H, then W: the pass over the full-size gradient is the fully coalesced one)
// read 1 + 1/2 + 1/4 and write 1/2 + 1/4 + 1/8 of the gradient tensor, with <= 5 taps per element — the one-pass gather above
// reads ~90 candidates per input voxel through L1/L2 (212 us on the full-resolution gradient at B=4; this: 3 bandwidth passes).
template <int V>  // V consecutive inner elements per thread (V = 4: 16-byte loads / stores; needs inner % 4 == 0)
__global__ void __launch_bounds__(256)
upsample2_adjoint_axis_kernel(const float *__restrict__ gout, float *__restrict__ gin, size_t outer, int in_len, size_t inner,
                              float scale) {
    const size_t inner_v = inner / V;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= outer * in_len * inner_v) return;
    const size_t q = (t % inner_v) * V;
    const int i = (int)((t / inner_v) % in_len);
    const size_t o_ = t / (inner_v * in_len);
    const AdjTaps tp = adj_taps(i, in_len, scale, 2 * in_len);
    const float *p = gout + (o_ * (size_t)(2 * in_len) + tp.lo) * inner + q;
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if (tp.w[k] == 0.0f) continue;
        if (V == 4) {
            const float4 g = *reinterpret_cast<const float4 *>(p + (size_t)k * inner);
            acc[0] += tp.w[k] * g.x; acc[1] += tp.w[k] * g.y; acc[2] += tp.w[k] * g.z; acc[3] += tp.w[k] * g.w;
        } else {
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += tp.w[k] * p[(size_t)k * inner + e];
        }
    }
    float *o = gin + (o_ * (size_t)in_len + i) * inner + q;
    if (V == 4) {
        *reinterpret_cast<float4 *>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
#pragma unroll
        for (int e = 0; e < V; ++e) o[e] = acc[e];
    }
}

static void launch_adjoint_axis(const float *gout, float *gin, size_t outer, int in_len, size_t inner, float scale, hipStream_t s) {
    const bool v4 = inner % 4 == 0 && ((uintptr_t)gout & 15) == 0 && ((uintptr_t)gin & 15) == 0;
    const size_t n = outer * in_len * (v4 ? inner / 4 : inner);
    if (v4)
        hipLaunchKernelGGL(upsample2_adjoint_axis_kernel<4>, dim3(cdiv(n, 256)), dim3(256), 0, s, gout, gin, outer, in_len, inner, scale);
    else
        hipLaunchKernelGGL(upsample2_adjoint_axis_kernel<1>, dim3(cdiv(n, 256)), dim3(256), 0, s, gout, gin, outer, in_len, inner, scale);
}

// nn.Upsample(scale_factor=(sD,sH,sW)) nearest backward: dx[i] = sum of the sD*sH*sW replicated outputs (model.py:427-433)
__global__ void __launch_bounds__(256)
upsample_nearest_bwd_kernel(const float *__restrict__ dout, float *__restrict__ dx, int D, int H, int W, int sD, int sH, int sW,
                            size_t total) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // over dx elements
    if (t >= total) return;
    const int w = (int)(t % W);
    size_t r = t / W;
    const int h = (int)(r % H);
    r /= H;
    const int d = (int)(r % D);
    const size_t plane = r / D;
    const int oH = H * sH, oW = W * sW;
    const float *p = dout + ((plane * (size_t)(D * sD) + (size_t)d * sD) * oH + (size_t)h * sH) * oW + (size_t)w * sW;
    float acc = 0.0f;
    for (int a = 0; a < sD; ++a)
        for (int b = 0; b < sH; ++b)
            for (int c = 0; c < sW; ++c) acc += p[((size_t)a * oH + b) * oW + c];
    dx[t] = acc;
}

// out[m][n] = sum_k (A[m*sam + k*sak] (+ A2[...])) * B[k*sbk + n*sbn]  (+ bias[n]): the tiny dense products of the
// warp generators' heads and their gradients ((z+e) @ Gamma, the 1x1 Conv2d on a 1x1 map; model.py:945-957, 446).
// One thread per output, double accumulation; M*N <= a few 10^6, K <= 2048.
__global__ void __launch_bounds__(256)
small_gemm_kernel(const float *__restrict__ A, const float *__restrict__ A2, const float *__restrict__ Bm,
                  const float *__restrict__ bias, float *__restrict__ out, int M, int N, int K, long sam, long sak, long sbk,
                  long sbn) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)M * N) return;
    const int n = (int)(t % N), m = (int)(t / N);
    double acc = bias ? (double)bias[n] : 0.0;
    for (int k = 0; k < K; ++k) {
        float a = A[m * sam + k * sak];
        if (A2) a += A2[m * sam + k * sak];
        acc += (double)a * (double)Bm[k * sbk + n * sbn];
    }
    out[t] = (float)acc;
}

// same product, one wavefront per output: the K loop is spread over the 64 lanes (few outputs, long K: ds = dx @ W)
__global__ void __launch_bounds__(256)
small_gemm_wave_kernel(const float *__restrict__ A, const float *__restrict__ A2, const float *__restrict__ Bm,
                       const float *__restrict__ bias, float *__restrict__ out, int M, int N, int K, long sam, long sak, long sbk,
                       long sbn) {
    const size_t t = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= (size_t)M * N) return;
    const int n = (int)(t % N), m = (int)(t / N), lane = threadIdx.x & 63;
    double acc = 0.0;
    for (int k = lane; k < K; k += 64) {
        float a = A[m * sam + k * sak];
        if (A2) a += A2[m * sam + k * sak];
        acc += (double)a * (double)Bm[k * sbk + n * sbn];
    }
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) acc += __shfl_xor(acc, sft, 64);
    if (lane == 0) out[t] = (float)(acc + (bias ? (double)bias[n] : 0.0));
}

// conv3d backward-weight for tiny volumes (FlowField's first blocks: 4..256 voxels per sample, up to 512x256x27
// outputs): one thread per dW element, looping over the voxels — the MFMA kernels above would spend their time
// staging mostly-padding tiles.  Exact fp32, fixed order.
template <int KS>
__global__ void __launch_bounds__(256)
conv_bwd_weight_direct_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ dw, int N, int Ci,
                              int Co, int D, int H, int W) {
    constexpr int TAPS = KS * KS * KS;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)Co * Ci * TAPS) return;
    const int tap = (int)(t % TAPS);
    const int ci = (int)((t / TAPS) % Ci), co = (int)(t / ((size_t)TAPS * Ci));
    const int kd = KS == 3 ? tap / 9 - 1 : 0, kh = KS == 3 ? (tap / 3) % 3 - 1 : 0, kw = KS == 3 ? tap % 3 - 1 : 0;
    const int HW = H * W, DHW = D * HW;
    // output voxels whose shifted input voxel is inside the volume
    const int d_lo = max(0, -kd), d_hi = min(D, D - kd), h_lo = max(0, -kh), h_hi = min(H, H - kh);
    const int w_lo = max(0, -kw), w_hi = min(W, W - kw);
    float acc = 0.0f;
    for (int n = 0; n < N; ++n) {
        const float *gy = dy + ((size_t)n * Co + co) * DHW;
        const float *xx = x + ((size_t)n * Ci + ci) * DHW + kd * HW + kh * W + kw;
        for (int d = d_lo; d < d_hi; ++d)
            for (int h = h_lo; h < h_hi; ++h)
                for (int w = w_lo; w < w_hi; ++w) acc += gy[d * HW + h * W + w] * xx[d * HW + h * W + w];
    }
    dw[t] = acc;
}

// small volumes (<= 4096 voxels over the batch: FlowField's middle blocks, 4x4 and 8x8 maps): one wavefront per
// (co, ci) pair, lanes over the voxels, all k^3 taps accumulated per lane and wave-reduced.  Exact fp32.
template <int KS>
__global__ void __launch_bounds__(256)
conv_bwd_weight_wave_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ dw, int N, int Ci,
                            int Co, int D, int H, int W) {
    constexpr int TAPS = KS * KS * KS;
    const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= Co * Ci) return;
    const int ci = pair % Ci, co = pair / Ci, lane = threadIdx.x & 63;
    const int HW = H * W, DHW = D * HW;
    float acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) acc[t] = 0.0f;
    for (int v = lane; v < N * DHW; v += 64) {
        const int n = v / DHW, r = v - n * DHW;
        const int d = r / HW, h = (r / W) % H, w = r % W;
        const float g = dy[((size_t)n * Co + co) * DHW + r];
        const float *xc = x + ((size_t)n * Ci + ci) * DHW + r;
        if (KS == 1) {
            acc[0] += g * xc[0];
        } else {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                if ((unsigned)(d + a - 1) >= (unsigned)D) continue;
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    if ((unsigned)(h + b - 1) >= (unsigned)H) continue;
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        if ((unsigned)(w + c - 1) < (unsigned)W) acc[(a * 3 + b) * 3 + c] += g * xc[(a - 1) * HW + (b - 1) * W + (c - 1)];
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        float a = acc[t];
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) a += __shfl_xor(a, sft, 64);
        if (lane == 0) dw[(size_t)pair * TAPS + t] = a;
    }
}

// The same small volumes as a GEMM on the exact-fp32 matrix cores: dW[co][ci][tap] = sum_v dY[co][v] * X[ci][v + tap] with
// M = 32 output channels, N = 32 input channels, K = the voxels of the batch, one tap per workgroup (v_mfma_f32_32x32x2_f32:
// lane l feeds row / column l & 31 and k index l >> 5).  A lane's K range is walked four voxels at a time: ONE 16-byte load of
// dY (its channel's row) and four 4-byte loads of the shifted X row serve four k-steps; voxels whose shifted neighbour is
// outside the volume contribute the zero padding.  The four waves of a workgroup split K and are folded through LDS in a fixed
// order: deterministic.  (The wave-per-pair kernel above spends 162 shuffles per 27 outputs: 48 us per FlowField layer.)
template <int KS, int NW>  // NW waves per workgroup split K (k = 1 has one tap, i.e. few workgroups: 16 waves each)
__global__ void __launch_bounds__(NW * 64)
conv_bwd_weight_small_mfma_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ dw, int N, int Ci,
                                  int Co, int D, int H, int W) {
    constexpr int TAPS = KS * KS * KS;
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    const int ci_tiles = (Ci + 31) / 32;
    const int co0 = (blockIdx.x / ci_tiles) * 32, ci0 = (blockIdx.x % ci_tiles) * 32;
    const int tap = blockIdx.y;
    const int kd = KS == 3 ? tap / 9 - 1 : 0, kh = KS == 3 ? (tap / 3) % 3 - 1 : 0, kw = KS == 3 ? tap % 3 - 1 : 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int HW = H * W, DHW = D * HW;
    const int co = min(co0 + col, Co - 1), ci = min(ci0 + col, Ci - 1);  // clamped rows: their products are never stored
    const bool ci_ok = ci0 + col < Ci;
    const int toff = kd * HW + kh * W + kw;
    // groups of 8 voxels inside one sample (DHW % 4 == 0 is required by the dispatcher): half h takes voxels 4h .. 4h+3 of a group
    const int groups_per_n = (DHW + 7) / 8, ngroups = N * groups_per_n;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    // branch-free loads (clamped addresses, validity applied as selects when the values are used), the next group's issued
    // before this group's MFMAs
    float4 a_cur, a_nxt;
    float b_cur[4], b_nxt[4];
    unsigned ok_cur = 0, ok_nxt = 0;  // bit e: voxel e exists and its shifted neighbour is inside the volume; bit 4: the group exists
    auto load = [&](int g, float4 &a4, float *b, unsigned &ok) {
        const int gc = min(g, ngroups - 1);
        const int n = gc / groups_per_n, r0 = min((gc % groups_per_n) * 8 + 4 * half, DHW - 4);
        const bool have = g < ngroups && (gc % groups_per_n) * 8 + 4 * half < DHW;
        a4 = *reinterpret_cast<const float4 *>(dy + ((size_t)n * Co + co) * DHW + r0);
        const float *xc = x + ((size_t)n * Ci + ci) * DHW;
        ok = have ? 16u : 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = r0 + e;
            const int d = r / HW, h = (r / W) % H, w = r % W;
            const bool in = (unsigned)(d + kd) < (unsigned)D && (unsigned)(h + kh) < (unsigned)H && (unsigned)(w + kw) < (unsigned)W;
            b[e] = xc[in ? r + toff : r];
            ok |= (have && in) ? (1u << e) : 0u;
        }
    };
    load(wave, a_cur, b_cur, ok_cur);
    for (int g = wave; g < ngroups; g += NW) {
        load(g + NW, a_nxt, b_nxt, ok_nxt);
        const float a[4] = {a_cur.x, a_cur.y, a_cur.z, a_cur.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32((ok_cur & 16u) ? a[e] : 0.0f, ((ok_cur >> e) & 1u) ? b_cur[e] : 0.0f, acc, 0, 0, 0);
        a_cur = a_nxt;
#pragma unroll
        for (int e = 0; e < 4; ++e) b_cur[e] = b_nxt[e];
        ok_cur = ok_nxt;
    }
    // fold the waves (plain LDS stores / loads, fixed order), then wave 0 writes the 32 x 32 block of this tap
    __shared__ float xch[NW - 1][16][64];
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[r];
            for (int q = 0; q < NW - 1; ++q) v += xch[q][r][lane];
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;  // D layout: row = output channel, column = lane & 31 = input channel
            if (co0 + row < Co && ci_ok) dw[((size_t)(co0 + row) * Ci + ci) * TAPS + tap] = v;
        }
    }
}

}  // namespace mphip

using namespace mphip;

// tiny volumes: one thread per dW element beats tiling (see conv_bwd_weight_direct_kernel)
static bool bwd_weight_direct(int N, int D, int H, int W) {
    const long vox = (long)N * D * H * W;
    return vox <= 64 || (W % 8 != 0 && vox <= 4096);  // maps narrower than the MFMA kernels' 8-wide voxel rows
}

static int bw_splits(long ntiles, int blocks_xy) {
    int s = 1;
    while ((long)blocks_xy * s < 1024 && ntiles / (s * 2) >= 8) s *= 2;
    return s;
}

extern "C" int mphip_conv3d_bwd_weight_supported(int N, int Ci, int Co, int D, int H, int W, int k, int precision) {
    if (N <= 0 || Ci <= 0 || Co <= 0 || D <= 0 || H <= 0 || W <= 0 || (k != 1 && k != 3)) return 0;
    if (precision == 0) return 1;
    return precision == 1 && bwd_weight_f16x3_supported(N, Ci, Co, D, H, W, k);
}

constexpr size_t BW_RANGE_BYTES = ((MPHIP_RANGE_FLOATS * sizeof(float) + 255) / 256) * 256;

extern "C" size_t mphip_conv3d_bwd_weight_workspace_bytes(int N, int Ci, int Co, int D, int H, int W, int k, int precision) {
    if (!mphip_conv3d_bwd_weight_supported(N, Ci, Co, D, H, W, k, precision)) return 0;
    if (bwd_weight_direct(N, D, H, W)) return 16;  // unused
    if (precision == 1) return BW_RANGE_BYTES + bwd_weight_f16x3_ws_bytes(N, Ci, Co, D, H, W, k);  // head: a library-computed range of x
    const long ntiles = (long)N * D * ((H + 7) / 8) * ((W + 7) / 8);
    const int bxy = ((Ci + 31) / 32) * ((Co + 95) / 96) * (k == 3 ? 3 : 1);
    return (size_t)bw_splits(ntiles, bxy) * Co * Ci * k * k * k * sizeof(float);
}

static int bwd_weight_impl(const float *x, const float *x_range, const float *dy, const float *dy_scale, float *dw, int N, int Ci, int Co, int D,
                           int H, int W, int k, int precision, void *workspace, size_t workspace_bytes, void *stream, const int *dy_boxes);

extern "C" int mphip_conv3d_bwd_weight(const float *x, const float *x_range, const float *dy, const float *dy_scale, float *dw, int N,
                                       int Ci, int Co, int D, int H, int W, int k, int precision, void *workspace,
                                       size_t workspace_bytes, void *stream) {
    return bwd_weight_impl(x, x_range, dy, dy_scale, dw, N, Ci, Co, D, H, W, k, precision, workspace, workspace_bytes, stream, nullptr);
}

// dW when dy is zero outside per-frame boxes {lx,ly,lz,ex,ey,ez,-,-} (the gradient of a gather): voxel tiles outside them are skipped
// by the f16x3 3x3x3 kernel (other kernels ignore the hint and read everything: same result).
extern "C" int mphip_conv3d_bwd_weight_roi(const float *x, const float *x_range, const float *dy, const float *dy_scale, float *dw,
                                           const int *dy_boxes, int N, int Ci, int Co, int D, int H, int W, int k, int precision,
                                           void *workspace, size_t workspace_bytes, void *stream) {
    MPHIP_REQUIRE(dy_boxes, "conv3d_bwd_weight_roi: null box list");
    return bwd_weight_impl(x, x_range, dy, dy_scale, dw, N, Ci, Co, D, H, W, k, precision, workspace, workspace_bytes, stream, dy_boxes);
}

static int bwd_weight_impl(const float *x, const float *x_range, const float *dy, const float *dy_scale, float *dw, int N, int Ci, int Co, int D,
                           int H, int W, int k, int precision, void *workspace, size_t workspace_bytes, void *stream, const int *dy_boxes) {
    MPHIP_REQUIRE(x && dy && dw, "conv3d_bwd_weight: null pointer");
    MPHIP_REQUIRE(N > 0 && Ci > 0 && Co > 0 && D > 0 && H > 0 && W > 0 && (k == 1 || k == 3), "conv3d_bwd_weight: bad dims");
    MPHIP_REQUIRE(mphip_conv3d_bwd_weight_supported(N, Ci, Co, D, H, W, k, precision),
                  "conv3d_bwd_weight: precision %d not available for this shape (query mphip_conv3d_bwd_weight_supported)", precision);
    MPHIP_REQUIRE(precision == 0 || dy_scale, "conv3d_bwd_weight: the f16x3 kernel needs the gradient scale of mphip_grad_prep");
    const size_t x_bytes = (size_t)N * Ci * D * H * W * sizeof(float);
    MPHIP_REQUIRE(x_bytes < 0x80000000ull, "conv3d_bwd_weight: input exceeds the 2 GiB buffer-addressing limit");
    const size_t need = mphip_conv3d_bwd_weight_workspace_bytes(N, Ci, Co, D, H, W, k, precision);
    if (!workspace || workspace_bytes < need) {
        set_error("conv3d_bwd_weight: workspace %zu bytes < required %zu", workspace_bytes, need);
        return MPHIP_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    if (bwd_weight_direct(N, D, H, W)) {
        const size_t nw = (size_t)Co * Ci * k * k * k;
        // a handful of voxels whose count the MFMA kernel's 16-byte loads cannot take: one thread per dW element (maps of 4 k
        // voxels, FlowField's 4x1x1 included, go to the MFMA kernel: 512x256x27 outputs 45 -> ~15 us)
        const bool per_thread = (long)N * D * H * W <= 64 && (D * H * W) % 4 != 0;
        if (per_thread && k == 3)
            hipLaunchKernelGGL(conv_bwd_weight_direct_kernel<3>, dim3(cdiv(nw, 256)), dim3(256), 0, s, x, dy, dw, N, Ci, Co, D, H, W);
        else if (per_thread)
            hipLaunchKernelGGL(conv_bwd_weight_direct_kernel<1>, dim3(cdiv(nw, 256)), dim3(256), 0, s, x, dy, dw, N, Ci, Co, D, H, W);
        else if ((D * H * W) % 4 == 0 && ((uintptr_t)dy & 15) == 0 && !getenv("MPHIP_BWD_WEIGHT_WAVE")) {  // (env: the older kernel, for A/B)
            const dim3 grid((unsigned)(((Co + 31) / 32) * ((Ci + 31) / 32)), k == 3 ? 27 : 1);
            if (k == 3)
                hipLaunchKernelGGL((conv_bwd_weight_small_mfma_kernel<3, 4>), grid, dim3(256), 0, s, x, dy, dw, N, Ci, Co, D, H, W);
            else
                hipLaunchKernelGGL((conv_bwd_weight_small_mfma_kernel<1, 16>), grid, dim3(1024), 0, s, x, dy, dw, N, Ci, Co, D, H, W);
        } else if (k == 3)
            hipLaunchKernelGGL(conv_bwd_weight_wave_kernel<3>, dim3(cdiv((size_t)Co * Ci, 4)), dim3(256), 0, s, x, dy, dw, N, Ci, Co, D,
                               H, W);
        else
            hipLaunchKernelGGL(conv_bwd_weight_wave_kernel<1>, dim3(cdiv((size_t)Co * Ci, 4)), dim3(256), 0, s, x, dy, dw, N, Ci, Co, D,
                               H, W);
        return check_launch("conv3d_bwd_weight(direct)");
    }
    if (precision == 1) {
        // the saved activation's range descriptor (the forward conv's x_range); none given -> computed into the workspace head
        if (!x_range) {
            int rc0 = absmax_range_launch(x, (size_t)N * Ci * D * H * W, (float *)workspace, s);
            if (rc0) return rc0;
            x_range = (const float *)workspace;
        }
        return bwd_weight_f16x3_launch(x, x_range, dy, dy_scale, dw, N, Ci, Co, D, H, W, k, (char *)workspace + BW_RANGE_BYTES, s,
                                       k == 3 ? dy_boxes : nullptr);
    }
    const long ntiles = (long)N * D * ((H + 7) / 8) * ((W + 7) / 8);
    const int ci_tiles = (Ci + 31) / 32, co_tiles = (Co + 95) / 96;
    const int splits = bw_splits(ntiles, ci_tiles * co_tiles * (k == 3 ? 3 : 1));
    const int tps = (int)((ntiles + splits - 1) / splits);
    dim3 grid(ci_tiles * co_tiles, k == 3 ? 3 : 1, splits);
    if (k == 3)
        hipLaunchKernelGGL(conv_bwd_weight_kernel<3>, grid, dim3(192), 0, s, x, dy, (float *)workspace, N, Ci, Co, D, H, W, tps,
                           (unsigned)x_bytes);
    else
        hipLaunchKernelGGL(conv_bwd_weight_kernel<1>, grid, dim3(192), 0, s, x, dy, (float *)workspace, N, Ci, Co, D, H, W, tps,
                           (unsigned)x_bytes);
    const size_t nw = (size_t)Co * Ci * k * k * k;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv(nw, 256)), dim3(256), 0, s, (const float *)workspace, dw, nw, splits);
    return check_launch("conv3d_bwd_weight");
}

extern "C" size_t mphip_groupnorm_bwd_workspace_bytes(int N, int C, int S) {
    return N > 0 && C > 0 && S > 0 ? ((size_t)N * C * cdiv(S, GNB_CHUNK) * 2 + (size_t)N * C * 2) * sizeof(float) : 0;
}

extern "C" int mphip_groupnorm_bwd_reduce(const float *x, const float *y, const float *dy, const float *stats,
                                          const float *gamma, const float *beta, const float *w2, float *dgamma, float *dbeta,
                                          float *dw2, float *db2, float *ab, int N, int C, int S, int G, int act,
                                          void *workspace, size_t workspace_bytes, void *stream) {
    const int relu = act;
    MPHIP_REQUIRE(x && dy && stats && gamma && dgamma && dbeta && ab && (!act || y), "groupnorm_bwd_reduce: null pointer");
    MPHIP_REQUIRE(!w2 || (beta && dw2 && db2), "groupnorm_bwd_reduce: the second affine needs beta, dw2 and db2");
    MPHIP_REQUIRE(act >= 0 && act <= 2, "groupnorm_bwd_reduce: act must be 0 (none), 1 (ReLU) or 2 (tanh(ReLU))");
    MPHIP_REQUIRE(N > 0 && C > 0 && S > 0 && G > 0 && C % G == 0, "groupnorm_bwd_reduce: bad dims");
    const size_t need = mphip_groupnorm_bwd_workspace_bytes(N, C, S);
    if (!workspace || workspace_bytes < need) {
        set_error("groupnorm_bwd_reduce: workspace %zu bytes < required %zu", workspace_bytes, need);
        return MPHIP_EWORKSPACE;
    }
    const int chunks = cdiv(S, GNB_CHUNK);
    float *partial = (float *)workspace, *s12 = partial + (size_t)N * C * chunks * 2;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(N * C * chunks), dim3(256), 0, s, x, y, dy, stats, partial, C, C / G, S, relu,
                       chunks);
    hipLaunchKernelGGL(gn_bwd_fold_kernel, dim3(1), dim3(1024), 0, s, (const float *)partial, gamma, beta, w2, s12, dgamma, dbeta,
                       dw2, db2, ab, N, C, C / G, S, chunks);
    return check_launch("groupnorm_bwd_reduce");
}

extern "C" int mphip_groupnorm_bwd_apply(const float *x, const float *y, const float *dy, const float *stats,
                                         const float *gamma, const float *w2, const float *ab, float *dx, float *dres, int N,
                                         int C, int S, int G, int act, void *stream) {
    const int relu = act;
    MPHIP_REQUIRE(x && dy && stats && gamma && ab && dx && (!act || y), "groupnorm_bwd_apply: null pointer");
    MPHIP_REQUIRE(N > 0 && C > 0 && S > 0 && G > 0 && C % G == 0, "groupnorm_bwd_apply: bad dims");
    const size_t total = (size_t)N * C * S;
    const bool aligned = (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)(y ? y : x) | (uintptr_t)(dres ? dres : dx)) & 15) == 0;
    if (S % 4 == 0 && S >= 512 && aligned && (size_t)N * C <= 65535)   // (tiny planes: the flat kernel fills its workgroups better)
        hipLaunchKernelGGL(gn_bwd_apply4_kernel, dim3((unsigned)cdiv(S, 1024), (unsigned)(N * C)), dim3(256), 0, (hipStream_t)stream, x, y, dy,
                           stats, gamma, w2, ab, dx, dres, C, C / G, S, relu);
    else
        hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y, dy, stats, gamma,
                           w2, ab, dx, dres, C, C / G, S, relu, total);
    return check_launch("groupnorm_bwd_apply");
}

extern "C" int mphip_groupnorm_bwd(const float *x, const float *y, const float *dy, const float *stats, const float *gamma,
                                   const float *beta, const float *w2, float *dx, float *dres, float *dgamma, float *dbeta, float *dw2,
                                   float *db2, int N, int C, int S, int G, int act, void *workspace, size_t workspace_bytes, void *stream) {
    const int relu = act;
    MPHIP_REQUIRE(x && dy && stats && gamma && dx && dgamma && dbeta && (!act || y), "groupnorm_bwd: null pointer");
    MPHIP_REQUIRE(!w2 || (beta && dw2 && db2), "groupnorm_bwd: the second affine needs beta, dw2 and db2");
    MPHIP_REQUIRE(act >= 0 && act <= 2, "groupnorm_bwd: act must be 0 (none), 1 (ReLU) or 2 (tanh(ReLU))");
    MPHIP_REQUIRE(N > 0 && C > 0 && S > 0 && G > 0 && C % G == 0, "groupnorm_bwd: bad dims");
    const size_t need = mphip_groupnorm_bwd_workspace_bytes(N, C, S);
    if (!workspace || workspace_bytes < need) {
        set_error("groupnorm_bwd: workspace %zu bytes < required %zu", workspace_bytes, need);
        return MPHIP_EWORKSPACE;
    }
    const int chunks = cdiv(S, GNB_CHUNK);
    float *partial = (float *)workspace;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(N * C * chunks), dim3(256), 0, s, x, y, dy, stats, partial, C, C / G, S, relu, chunks);
    const size_t total = (size_t)N * C * S;
    const bool aligned = (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)(y ? y : x) | (uintptr_t)(dres ? dres : dx)) & 15) == 0;
    if (S % 4 == 0 && S >= 512 && aligned && (size_t)N * C <= 65535)
        hipLaunchKernelGGL(gn_bwd_apply4_fold_kernel, dim3((unsigned)cdiv(S, 1024), (unsigned)(N * C)), dim3(256), 0, s, x, y, dy, stats, gamma,
                           beta, w2, (const float *)partial, dx, dres, dgamma, dbeta, dw2, db2, N, C, C / G, S, relu, chunks);
    else
        hipLaunchKernelGGL(gn_bwd_apply_fold_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, x, y, dy, stats, gamma, beta, w2,
                           (const float *)partial, dx, dres, dgamma, dbeta, dw2, db2, N, C, C / G, S, relu, chunks, total);
    return check_launch("groupnorm_bwd");
}

extern "C" int mphip_avgpool2_bwd(const float *dout, float *dx, int NC, int D, int H, int W, void *stream) {
    MPHIP_REQUIRE(dout && dx, "avgpool2_bwd: null pointer");
    MPHIP_REQUIRE(NC > 0 && D > 0 && H > 0 && W > 0 && D % 2 == 0 && H % 2 == 0 && W % 2 == 0, "avgpool2_bwd: bad dims");
    const size_t total = (size_t)NC * D * H * W;
    hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, dout, dx, D, H, W, total);
    return check_launch("avgpool2_bwd");
}

extern "C" size_t mphip_upsample_trilinear2_bwd_workspace_bytes(int NC, int D, int H, int W) {
    if (NC <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)NC * D * H * W * (4 + 2) * sizeof(float);  // [NC,D,2H,2W] and [NC,D,H,2W]
}

extern "C" int mphip_upsample_trilinear2_bwd(const float *dout, float *dx, int NC, int D, int H, int W, void *workspace,
                                             size_t workspace_bytes, void *stream) {
    MPHIP_REQUIRE(dout && dx, "upsample_trilinear2_bwd: null pointer");
    MPHIP_REQUIRE(NC > 0 && D > 0 && H > 0 && W > 0, "upsample_trilinear2_bwd: bad dims");
    const float sD = 2 * D > 1 ? (float)(D - 1) / (float)(2 * D - 1) : 0.0f;
    const float sH = 2 * H > 1 ? (float)(H - 1) / (float)(2 * H - 1) : 0.0f;
    const float sW = 2 * W > 1 ? (float)(W - 1) / (float)(2 * W - 1) : 0.0f;
    const size_t total = (size_t)NC * D * H * W;
    hipStream_t s = (hipStream_t)stream;
    if (!workspace) {  // no scratch: the one-pass gather
        hipLaunchKernelGGL(upsample_trilinear2_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, dout, dx, D, H, W, sD, sH, sW, total);
        return check_launch("upsample_trilinear2_bwd");
    }
    const size_t need = mphip_upsample_trilinear2_bwd_workspace_bytes(NC, D, H, W);
    if (workspace_bytes < need) {
        set_error("upsample_trilinear2_bwd: workspace %zu bytes < required %zu", workspace_bytes, need);
        return MPHIP_EWORKSPACE;
    }
    float *t1 = (float *)workspace, *t2 = t1 + total * 4;
    launch_adjoint_axis(dout, t1, (size_t)NC, D, (size_t)4 * H * W, sD, s);
    launch_adjoint_axis(t1, t2, (size_t)NC * D, H, (size_t)2 * W, sH, s);
    launch_adjoint_axis(t2, dx, (size_t)NC * D * H, W, (size_t)1, sW, s);
    return check_launch("upsample_trilinear2_bwd");
}

// adjoint of mphip_upsample_trilinear: every output gradient scatters to its 8 source voxels with the forward's weights
// (hardware fp32 atomics, like ATen's upsample_trilinear3d backward on the GPU)
__global__ void __launch_bounds__(256) upsample_trilinear_scaled_bwd_kernel(const float *__restrict__ dout, float *__restrict__ dx,
                                                                            int D, int H, int W, int sD, int sH, int sW, size_t total) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int oD = D * sD, oH = H * sH, oW = W * sW;
    int ow = (int)(t % oW);
    size_t r = t / oW;
    int oh = (int)(r % oH);
    r /= oH;
    int od = (int)(r % oD);
    size_t plane = r / oD;
    const SrcIdx sd = src_index<false>(od, D, oD), sh = src_index<false>(oh, H, oH), sw = src_index<false>(ow, W, oW);
    const float g = dout[t];
    float *p = dx + plane * D * H * W;
    const int di[2] = {sd.i0, sd.i1}, hi[2] = {sh.i0, sh.i1}, wi[2] = {sw.i0, sw.i1};
    const float dl[2] = {sd.l0, sd.l1}, hl[2] = {sh.l0, sh.l1}, wl[2] = {sw.l0, sw.l1};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float wgt = dl[a] * hl[b] * wl[c];
                if (wgt != 0.0f) atomicAdd(p + ((size_t)di[a] * H + hi[b]) * W + wi[c], g * wgt);
            }
}

extern "C" int mphip_upsample_trilinear_bwd(const float *dout, float *dx, int NC, int D, int H, int W, int sD, int sH, int sW,
                                            void *stream) {
    MPHIP_REQUIRE(dout && dx, "upsample_trilinear_bwd: null pointer");
    MPHIP_REQUIRE(NC > 0 && D > 0 && H > 0 && W > 0 && sD > 0 && sH > 0 && sW > 0, "upsample_trilinear_bwd: bad dims");
    MPHIP_REQUIRE(((size_t)NC * D * H * W * sizeof(float)) % 16 == 0 && ((uintptr_t)dx & 15) == 0,
                  "upsample_trilinear_bwd: dx must be 16-byte aligned and a multiple of 16 bytes");
    const size_t total = (size_t)NC * D * H * W * sD * sH * sW;
    zero_fill(dx, (size_t)NC * D * H * W * sizeof(float), (hipStream_t)stream);
    hipLaunchKernelGGL(upsample_trilinear_scaled_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, dout, dx, D,
                       H, W, sD, sH, sW, total);
    return check_launch("upsample_trilinear_bwd");
}

extern "C" int mphip_upsample_nearest_bwd(const float *dout, float *dx, int NC, int D, int H, int W, int sD, int sH, int sW,
                                          void *stream) {
    MPHIP_REQUIRE(dout && dx, "upsample_nearest_bwd: null pointer");
    MPHIP_REQUIRE(NC > 0 && D > 0 && H > 0 && W > 0 && sD > 0 && sH > 0 && sW > 0, "upsample_nearest_bwd: bad dims");
    const size_t total = (size_t)NC * D * H * W;
    hipLaunchKernelGGL(upsample_nearest_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, dout, dx, D, H, W, sD,
                       sH, sW, total);
    return check_launch("upsample_nearest_bwd");
}

extern "C" int mphip_small_gemm(const float *a, const float *a2, const float *b, const float *bias, float *out, int M, int N,
                                int K, long sam, long sak, long sbk, long sbn, void *stream) {
    MPHIP_REQUIRE(a && b && out, "small_gemm: null pointer");
    MPHIP_REQUIRE(M > 0 && N > 0 && K > 0, "small_gemm: bad dims");
    if ((size_t)M * N <= 65536 && K >= 128)
        hipLaunchKernelGGL(small_gemm_wave_kernel, dim3(cdiv((size_t)M * N, 4)), dim3(256), 0, (hipStream_t)stream, a, a2, b, bias,
                           out, M, N, K, sam, sak, sbk, sbn);
    else
        hipLaunchKernelGGL(small_gemm_kernel, dim3(cdiv((size_t)M * N, 256)), dim3(256), 0, (hipStream_t)stream, a, a2, b, bias, out,
                           M, N, K, sam, sak, sbk, sbn);
    return check_launch("small_gemm");
}
