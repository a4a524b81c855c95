// K1b — the first three residual blocks of FlowField (reference model.py:369-408 at the fixed levels of model.py:439-471: 512->256 @4x1x1,
// 256->128 @8x2x2, 128->64 @16x4x4) as TWO launches per block instead of five:
//   A:  a   = relu(AGN1(conv1(x)))
//   B:  out = upsample_nearest(relu(AGN2(conv2(a)) + residual_conv(x)))
// These tensors are tiny (<= 256 voxels per frame) and the chain of ~25 dependent 5-30 us launches per generator was what the head of
// every step cost (DESIGN.md "Small-volume path").  The split-K gather convs needed a second launch per conv to fold their slabs before
// the GroupNorm could see a whole group; here ONE workgroup owns a whole (frame, GroupNorm group): it runs the full input-channel
// loop for the group's channels — sliced over its own 512-1024 threads and folded through LDS in slice order — so the statistics, both affines,
// the residual 1x1x1 conv, ReLU and the nearest upsample happen in the same launch.  32 groups x 8 frames = 256 workgroups = the chip.
// Arithmetic: fp32 FMA chains from the ORIGINAL [Co][Ci][27] weights (no packed copy to refresh after an optimizer step).
//
// Two thread mappings (all shapes are compile-time: the levels of FlowField do not depend on the image size):
//   PLANE (4x1x1, 8x2x2):  thread = (channel of the group, slice of the input channels); the whole input plane of one input channel
//                           lives in registers and every in-bounds (output voxel, tap) pair is one unrolled FMA.
//   ROW   (16x4x4):         thread = (one w-row of outputs, slice of the input channels); per input channel the 9 neighbouring rows are
//                           loaded once (16-byte loads) and feed 3 taps x W outputs x the group's channels; the weights of a slice
//                           are wave-uniform (scalar loads).
#include <stdlib.h>

#include "mphip_common.h"

namespace mphip {

struct FfParams {
    const float *x, *w, *b;                    // [N,Ci,S], [Co,Ci,27], [Co]
    const float *gamma, *beta, *w2, *b2;       // group_norm.{weight,bias}, AdaptiveGroupNorm.{weight,bias} (model.py:304-316)
    const float *rx, *rw, *rb;                 // residual 1x1x1 conv: input [N,Cr,S], weight [Co,Cr], bias [Co] (or all null)
    float *y;                                  // [N,Co,D*uD,H*uH,W*uW]
    int Ci, Co, Cr, uD, uH, uW, relu;
    float eps;
};

__device__ __forceinline__ double ff_wave_sum(double v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
    return v;
}

// GroupNorm over the group's CPG*S values in LDS (bias already added), both affines, residual, ReLU, nearest upsample -> y.
// Every thread of the 256 calls it (barriers inside).
template <int D, int H, int W, int CPG, int NT>
__device__ __forceinline__ void ff_finish(const FfParams &p, const float *__restrict__ vals, const float *__restrict__ res, int n, int c0,
                                          double *red /* [2 * NT / 64] */, float *mr /* [2] */) {
    constexpr int S = D * H * W, CNT = CPG * S;
    const int tid = threadIdx.x;
    float s = 0.0f, ss = 0.0f;
    for (int e = tid; e < CNT; e += NT) {
        const float v = vals[e];
        s += v;
        ss += v * v;
    }
    const double ds = ff_wave_sum((double)s), dss = ff_wave_sum((double)ss);
    if ((tid & 63) == 0) {
        red[(tid >> 6) * 2] = ds;
        red[(tid >> 6) * 2 + 1] = dss;
    }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0, b = 0.0;
        for (int wv = 0; wv < NT / 64; ++wv) {   // wave order: deterministic
            a += red[wv * 2];
            b += red[wv * 2 + 1];
        }
        const double mean = a / (double)CNT;
        double var = b / (double)CNT - mean * mean;
        if (var < 0.0) var = 0.0;
        mr[0] = (float)mean;
        mr[1] = (float)(1.0 / sqrt(var + (double)p.eps));
    }
    __syncthreads();
    const float mean = mr[0], rstd = mr[1];
    const int oH = H * p.uH, oW = W * p.uW, oD = D * p.uD;
    const size_t oS = (size_t)oD * oH * oW;
    for (int e = tid; e < CNT; e += NT) {
        const int c = e / S, i = e - c * S, ch = c0 + c;
        float v = (vals[e] - mean) * rstd * p.gamma[ch] + p.beta[ch];
        if (p.w2) v = v * p.w2[ch] + p.b2[ch];
        if (res) v += res[e] + (p.rb ? p.rb[ch] : 0.0f);
        if (p.relu) v = fmaxf(v, 0.0f);
        const int d = i / (H * W), h = (i / W) % H, w = i % W;
        float *dst = p.y + ((size_t)n * p.Co + ch) * oS;
        for (int a = 0; a < p.uD; ++a)
            for (int b = 0; b < p.uH; ++b)
                for (int cc = 0; cc < p.uW; ++cc) dst[((size_t)(d * p.uD + a) * oH + h * p.uH + b) * oW + w * p.uW + cc] = v;
    }
}

// Fold the KS per-slice partial results red_[(k*CPG + c)*S + o] into dst[c*S + o] (+ bias[c0 + c]): NT / (CPG*S) threads share an output
// (contiguous slice ranges, then the ranges in order — a fixed tree: deterministic).  Every thread calls it (barriers inside).
template <int CPG, int S, int KS, int NT, int SP = S>   // SP: floats between the rows of red_
__device__ __forceinline__ void ff_fold(const float *__restrict__ red_, float *__restrict__ part_, float *__restrict__ dst,
                                        const float *__restrict__ bias, int c0) {
    constexpr int OUT = CPG * S, PARTS = (NT / OUT) < 1 ? 1 : ((NT / OUT) > KS ? KS : (NT / OUT)), PER = KS / PARTS;
    static_assert(KS % PARTS == 0, "slices split evenly over the folding threads");
    const int tid = threadIdx.x;
    for (int t = tid; t < OUT * PARTS; t += NT) {
        const int e = t % OUT, q = t / OUT, cc = e / S, o = e - cc * S;
        float v = red_[((q * PER) * CPG + cc) * SP + o];
        for (int kk = q * PER + 1; kk < (q + 1) * PER; ++kk) v += red_[(kk * CPG + cc) * SP + o];
        part_[t] = v;
    }
    __syncthreads();
    for (int e = tid; e < OUT; e += NT) {
        float v = part_[e];
        for (int q = 1; q < PARTS; ++q) v += part_[q * OUT + e];
        dst[e] = v + (bias ? bias[c0 + e / S] : 0.0f);
    }
}

// ---- PLANE: thread = (channel c of the group, input-channel slice k); KS = NT / CPG slices (a slice is 1-4 input channels: the
// loop is a chain of dependent global-load round trips, so the workgroup is made as wide as the channel count allows) --------------------------------------
// WC (level 1 only, r04): p.w is the COMPACT copy [Co][Ci][3] of the three taps (kd, 1, 1) a 4x1x1 volume can use (mphip_flowfield_compact_weight).  From
// the [Co][Ci][27] tensor the kernel touches every 128-byte line of its group's rows for 12 of every 108 bytes: 113 MB of L2 reads per launch at B=8
// for 12.6 MB of weights, and that traffic IS the launch (16-22 us -> see DESIGN.md).
template <int D, int H, int W, int CPG, int NT, bool WC = false>
__global__ void __launch_bounds__(NT) ff_block_plane_kernel(FfParams p) {
    static_assert(!WC || (H == 1 && W == 1), "compact taps: only (kd, 1, 1) exist");
    constexpr int S = D * H * W, KS = NT / CPG;
    // rows of the per-slice partials are S + 4 floats apart: thread t writes row t, and with a pitch of S = 32 floats every lane of a store
    // hit the same banks (68 % of this kernel's LDS cycles were bank conflicts, tools/pmc_step_lds.sh); 36 floats spread eight 16-byte stores
    // over all 32 banks
    constexpr int SP = S + 4;
    static_assert(S % 4 == 0 && S <= 32, "plane mapping: the input plane lives in registers");
    const int tid = threadIdx.x, c = tid % CPG, k = tid / CPG;
    const int groups = p.Co / CPG, n = blockIdx.x / groups, g = blockIdx.x % groups, c0 = g * CPG;
    __shared__ __attribute__((aligned(16))) float red_[KS * CPG * SP];
    __shared__ float vals[CPG * S], resv[CPG * S], part_[NT > CPG * S ? NT : CPG * S];
    __shared__ double dred[2 * NT / 64];
    __shared__ float mr[2];

    float acc[S];
#pragma unroll
    for (int o = 0; o < S; ++o) acc[o] = 0.0f;
    const int per = p.Ci / KS;   // 1..4 input channels: unrolled, so that every load of the slice is in flight at once
#pragma unroll 4
    for (int ci = k * per; ci < (k + 1) * per; ++ci) {
        float xp[S], wv[27];
        const float4 *xs = reinterpret_cast<const float4 *>(p.x + ((size_t)n * p.Ci + ci) * S);
#pragma unroll
        for (int q = 0; q < S / 4; ++q) {
            const float4 v = xs[q];
            xp[q * 4] = v.x; xp[q * 4 + 1] = v.y; xp[q * 4 + 2] = v.z; xp[q * 4 + 3] = v.w;
        }
        if (WC) {
            const float *wp = p.w + ((size_t)(c0 + c) * p.Ci + ci) * 3;
#pragma unroll
            for (int t = 0; t < 27; ++t) wv[t] = 0.0f;
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) wv[kd * 9 + 4] = wp[kd];
        } else {
            const float *wp = p.w + ((size_t)(c0 + c) * p.Ci + ci) * 27;
#pragma unroll
            for (int t = 0; t < 27; ++t) wv[t] = wp[t];   // (taps that no output of this shape can use are never loaded: dead after unrolling)
        }
#pragma unroll
        for (int od = 0; od < D; ++od)
#pragma unroll
            for (int oh = 0; oh < H; ++oh)
#pragma unroll
                for (int ow = 0; ow < W; ++ow)
#pragma unroll
                    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                            for (int kw = 0; kw < 3; ++kw) {
                                const int id = od + kd - 1, ih = oh + kh - 1, iw = ow + kw - 1;
                                if (id >= 0 && id < D && ih >= 0 && ih < H && iw >= 0 && iw < W)
                                    acc[(od * H + oh) * W + ow] =
                                        __builtin_fmaf(wv[kd * 9 + kh * 3 + kw], xp[(id * H + ih) * W + iw], acc[(od * H + oh) * W + ow]);
                            }
    }
#pragma unroll
    for (int o = 0; o < S; ++o) red_[(k * CPG + c) * SP + o] = acc[o];
    __syncthreads();
    ff_fold<CPG, S, KS, NT, SP>(red_, part_, vals, p.b, c0);
    if (p.rx) {   // (uniform) residual 1x1x1 conv of the block input, same slicing
        __syncthreads();
#pragma unroll
        for (int o = 0; o < S; ++o) acc[o] = 0.0f;
        const int perr = p.Cr / KS;
#pragma unroll 4
        for (int cr = k * perr; cr < (k + 1) * perr; ++cr) {
            const float wr = p.rw[(size_t)(c0 + c) * p.Cr + cr];
            const float4 *xs = reinterpret_cast<const float4 *>(p.rx + ((size_t)n * p.Cr + cr) * S);
#pragma unroll
            for (int q = 0; q < S / 4; ++q) {
                const float4 v = xs[q];
                acc[q * 4] = __builtin_fmaf(wr, v.x, acc[q * 4]);
                acc[q * 4 + 1] = __builtin_fmaf(wr, v.y, acc[q * 4 + 1]);
                acc[q * 4 + 2] = __builtin_fmaf(wr, v.z, acc[q * 4 + 2]);
                acc[q * 4 + 3] = __builtin_fmaf(wr, v.w, acc[q * 4 + 3]);
            }
        }
#pragma unroll
        for (int o = 0; o < S; ++o) red_[(k * CPG + c) * SP + o] = acc[o];
        __syncthreads();
        ff_fold<CPG, S, KS, NT, SP>(red_, part_, resv, nullptr, c0);
    }
    __syncthreads();
    ff_finish<D, H, W, CPG, NT>(p, vals, p.rx ? resv : nullptr, n, c0, dred, mr);
}

// ---- ROW: thread = (output row (d,h), input-channel slice k); rows * KS = 1024 threads (four waves per SIMD cover the load latency) ----------------------------------------------
#ifdef MPHIP_FF_TRACE   /* dev: wall-clock (100 MHz) stamps per workgroup of the ROW kernel: start, conv loop, fold, residual loop, fold, end */
__device__ unsigned long long g_ff_trace[512 * 8];
#define FF_STAMP(i) if (threadIdx.x == 0 && blockIdx.x < 512) g_ff_trace[blockIdx.x * 8 + (i)] = wall_clock64();
#else
#define FF_STAMP(i)
#endif
template <int D, int H, int W, int CPG, int KS>
__global__ void __launch_bounds__(D * H * KS) ff_block_row_kernel(FfParams p) {
    FF_STAMP(0)
    constexpr int S = D * H * W, R = D * H, NT = R * KS;
    static_assert(NT <= 1024 && R % 64 == 0 && W % 4 == 0, "row mapping: a wave holds rows of ONE slice");
    const int tid = threadIdx.x, r = tid % R, d = r / H, h = r % H;
    const int k = __builtin_amdgcn_readfirstlane(tid / R);   // wave-uniform: the slice's weights are scalar loads
    const int groups = p.Co / CPG, n = blockIdx.x / groups, g = blockIdx.x % groups, c0 = g * CPG;
    __shared__ float red_[KS * CPG * S];
    __shared__ float vals[CPG * S], resv[CPG * S], part_[NT > CPG * S ? NT : CPG * S];
    __shared__ double dred[2 * NT / 64];
    __shared__ float mr[2];

    float acc[CPG][W];
#pragma unroll
    for (int c = 0; c < CPG; ++c)
#pragma unroll
        for (int ow = 0; ow < W; ++ow) acc[c][ow] = 0.0f;
    const int per = p.Ci / KS;
    for (int ci = k * per; ci < (k + 1) * per; ++ci) {
        const float *xc = p.x + ((size_t)n * p.Ci + ci) * S;
        float row[9][W];
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int id = d + kd - 1, ih = h + kh - 1;
                const bool ok = id >= 0 && id < D && ih >= 0 && ih < H;
                const float4 *src = reinterpret_cast<const float4 *>(xc + ((size_t)(ok ? id : 0) * H + (ok ? ih : 0)) * W);
#pragma unroll
                for (int q = 0; q < W / 4; ++q) {
                    const float4 v = ok ? src[q] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    row[kd * 3 + kh][q * 4] = v.x; row[kd * 3 + kh][q * 4 + 1] = v.y;
                    row[kd * 3 + kh][q * 4 + 2] = v.z; row[kd * 3 + kh][q * 4 + 3] = v.w;
                }
            }
#pragma unroll
        for (int c = 0; c < CPG; ++c) {
            const float *wp = p.w + ((size_t)(c0 + c) * p.Ci + ci) * 27;   // wave-uniform address
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float wt = wp[t9 * 3 + kw];
#pragma unroll
                    for (int ow = 0; ow < W; ++ow) {
                        const int iw = ow + kw - 1;
                        if (iw >= 0 && iw < W) acc[c][ow] = __builtin_fmaf(wt, row[t9][iw], acc[c][ow]);
                    }
                }
        }
    }
    FF_STAMP(1)
#pragma unroll
    for (int c = 0; c < CPG; ++c)
#pragma unroll
        for (int ow = 0; ow < W; ++ow) red_[(k * CPG + c) * S + r * W + ow] = acc[c][ow];
    __syncthreads();
    ff_fold<CPG, S, KS, NT>(red_, part_, vals, p.b, c0);
    FF_STAMP(2)
    if (p.rx) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CPG; ++c)
#pragma unroll
            for (int ow = 0; ow < W; ++ow) acc[c][ow] = 0.0f;
        const int perr = p.Cr / KS;
        for (int cr = k * perr; cr < (k + 1) * perr; ++cr) {
            const float4 *src = reinterpret_cast<const float4 *>(p.rx + ((size_t)n * p.Cr + cr) * S + (size_t)r * W);
            float xr[W];
#pragma unroll
            for (int q = 0; q < W / 4; ++q) {
                const float4 v = src[q];
                xr[q * 4] = v.x; xr[q * 4 + 1] = v.y; xr[q * 4 + 2] = v.z; xr[q * 4 + 3] = v.w;
            }
#pragma unroll
            for (int c = 0; c < CPG; ++c) {
                const float wr = p.rw[(size_t)(c0 + c) * p.Cr + cr];
#pragma unroll
                for (int ow = 0; ow < W; ++ow) acc[c][ow] = __builtin_fmaf(wr, xr[ow], acc[c][ow]);
            }
        }
#pragma unroll
        for (int c = 0; c < CPG; ++c)
#pragma unroll
            for (int ow = 0; ow < W; ++ow) red_[(k * CPG + c) * S + r * W + ow] = acc[c][ow];
        FF_STAMP(3)
        __syncthreads();
        ff_fold<CPG, S, KS, NT>(red_, part_, resv, nullptr, c0);
    }
    FF_STAMP(4)
    __syncthreads();
    ff_finish<D, H, W, CPG, NT>(p, vals, p.rx ? resv : nullptr, n, c0, dred, mr);
    FF_STAMP(5)
}

// ---- the output head of FlowField (model.py:458-465): Conv3d(32, 3, 3) @16x16x16 -> GroupNorm(1, 3) -> ReLU -> tanh ----------------------
// The 3-channel conv is 90 % padding on a 32-row MFMA tile (29-50 us as a split-K gather conv) and its GroupNorm has ONE group per frame
// (a one-workgroup-per-group kernel uses 8 CUs: 27-51 us).  Here: (1) a direct conv, thread = voxel, the three channels in registers,
// weights wave-uniform, one workgroup per depth slice with the slice's input staged in LDS -> y and per-slice (sum, sum of squares) in double; (2) every workgroup folds its
// frame's 16 slice partials in slice order (deterministic) and applies the norm, ReLU and tanh to its slice.
constexpr int FO_C = 32, FO_G = 16, FO_S = FO_G * FO_G * FO_G;

__global__ void __launch_bounds__(256) ff_out_conv_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ b,
                                                          float *__restrict__ y, double *__restrict__ part) {
    // the slice's three input planes of all 32 channels, zero-padded by one voxel in h and w (and whole planes of zeros outside the
    // volume in d): 32 x 3 x 18 x 18 floats = 124 KB of LDS, staged once; the 27 x 32 taps of a voxel are then plain LDS reads
    constexpr int P = FO_G + 2, PLANE = P * P, XS = FO_C * 3 * PLANE;
    __shared__ float xs[XS];
    __shared__ __attribute__((aligned(16))) float wsd[FO_C * 3 * 28];   // weights [ci][c][27 (+1 pad)]: 16-byte LDS reads, no scalar-load round
                                                                       // trip per input channel (one wave per SIMD: nothing would hide it)
    // a 32-lane read group holds rows g and g + 8 of the slice: their 18-float-pitch windows are 144 floats = 16 banks apart, so the two
    // 16-wide tap reads of a group fill the 32 banks exactly (rows g, g + 1 overlapped on two banks: 27 % of the LDS cycles were conflicts)
    const int n = blockIdx.x / FO_G, d = blockIdx.x % FO_G, tid = threadIdx.x, h = (tid >> 5) + 8 * ((tid >> 4) & 1), ww = tid % FO_G;
    const float *xn = x + (size_t)n * FO_C * FO_S;
    // every global load of the workgroup is issued before the first use: 24 16-byte loads of the slab + 3 of the weights per thread
    // (a loop of dependent load -> LDS-write trips cost one memory round trip each: 57-70 us for this kernel)
    float4 xv[24];
#pragma unroll
    for (int q = 0; q < 24; ++q) {
        const int f = q * 256 + tid;                     // float4 index in [ci][kd][h][w/4]
        const int w4 = f % 4, hh = (f / 4) % FO_G, kd = (f / 64) % 3, ci = f / 192, id = d + kd - 1;
        xv[q] = (id >= 0 && id < FO_G) ? *reinterpret_cast<const float4 *>(xn + (size_t)ci * FO_S + (id * FO_G + hh) * FO_G + w4 * 4)
                                        : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    float wl[11];
#pragma unroll
    for (int q = 0; q < 11; ++q) {
        const int i = q * 256 + tid;                     // [ci][c][28]
        const int t = i % 28, c = (i / 28) % 3, ci = i / 84;
        wl[q] = (i < FO_C * 3 * 28 && t < 27) ? w[((size_t)c * FO_C + ci) * 27 + t] : 0.0f;
    }
    for (int i = tid; i < FO_C * 3 * P; i += 256) {      // the padding: columns 0 and 17 of every row, rows 0 and 17 of every plane
        const int r = i % P, pl = i / P;
        xs[pl * PLANE + r * P] = 0.0f;
        xs[pl * PLANE + r * P + P - 1] = 0.0f;
        xs[pl * PLANE + r] = 0.0f;
        xs[pl * PLANE + (P - 1) * P + r] = 0.0f;
    }
#pragma unroll
    for (int q = 0; q < 24; ++q) {
        const int f = q * 256 + tid;
        const int w4 = f % 4, hh = (f / 4) % FO_G, pl = f / 64;   // pl = ci*3 + kd
        float *dst = xs + pl * PLANE + (hh + 1) * P + 1 + w4 * 4;
        dst[0] = xv[q].x; dst[1] = xv[q].y; dst[2] = xv[q].z; dst[3] = xv[q].w;
    }
#pragma unroll
    for (int q = 0; q < 11; ++q) {
        const int i = q * 256 + tid;
        if (i < FO_C * 3 * 28) wsd[i] = wl[q];
    }
    __syncthreads();
    float acc[3] = {0.0f, 0.0f, 0.0f};
    for (int ci = 0; ci < FO_C; ++ci) {
        const float *xc = xs + ci * 3 * PLANE + h * P + ww;   // tap (kd, kh, kw) at + kd*PLANE + kh*P + kw
        float v[27];
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) v[kd * 9 + kh * 3 + kw] = xc[kd * PLANE + kh * P + kw];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float4 *wp = reinterpret_cast<const float4 *>(wsd + (ci * 3 + c) * 28);
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                const float4 wq = wp[q];
                acc[c] = __builtin_fmaf(wq.x, v[q * 4], acc[c]);
                if (q * 4 + 1 < 27) acc[c] = __builtin_fmaf(wq.y, v[q * 4 + 1], acc[c]);
                if (q * 4 + 2 < 27) acc[c] = __builtin_fmaf(wq.z, v[q * 4 + 2], acc[c]);
                if (q * 4 + 3 < 27) acc[c] = __builtin_fmaf(wq.w, v[q * 4 + 3], acc[c]);
            }
        }
    }
    float s = 0.0f, ss = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float val = acc[c] + (b ? b[c] : 0.0f);
        y[((size_t)n * 3 + c) * FO_S + d * FO_G * FO_G + h * FO_G + ww] = val;
        s += val;
        ss += val * val;
    }
    const double ds = ff_wave_sum((double)s), dss = ff_wave_sum((double)ss);
    __shared__ double red[8];
    if ((tid & 63) == 0) {
        red[(tid >> 6) * 2] = ds;
        red[(tid >> 6) * 2 + 1] = dss;
    }
    __syncthreads();
    if (tid == 0) {
        part[(size_t)blockIdx.x * 2] = (red[0] + red[2]) + (red[4] + red[6]);
        part[(size_t)blockIdx.x * 2 + 1] = (red[1] + red[3]) + (red[5] + red[7]);
    }
}

__global__ void __launch_bounds__(256) ff_out_norm_kernel(const float *__restrict__ y, const double *__restrict__ part, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, float eps, float *__restrict__ em) {
    const int n = blockIdx.x / FO_G, d = blockIdx.x % FO_G, tid = threadIdx.x;
    double a = 0.0, q = 0.0;
    for (int t = 0; t < FO_G; ++t) {   // (every thread folds the same 16 pairs in the same order: uniform, no barrier)
        a += part[((size_t)n * FO_G + t) * 2];
        q += part[((size_t)n * FO_G + t) * 2 + 1];
    }
    const double cnt = 3.0 * FO_S, mean_d = a / cnt;
    double var = q / cnt - mean_d * mean_d;
    if (var < 0.0) var = 0.0;
    const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var + (double)eps));
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const size_t i = ((size_t)n * 3 + c) * FO_S + d * FO_G * FO_G + tid;
        float v = (y[i] - mean) * rstd * gamma[c] + beta[c];
        v = fmaxf(v, 0.0f);
        em[i] = tanhf(v);
    }
}

// level of FlowField a (Co, D, H, W) belongs to: 1..3, 0 = none.  (Level 4, 64->32 @16x8x8, stays on the split-K gather conv + one-launch
// GroupNorm: measured as a row-mapped kernel it took 29-35 us per half — every one of the 32 group workgroups of a frame re-reads the
// frame's whole 256 KB input through L2 and the fp32 FMA work alone is 11 us of a CU — vs 107 us for its five launches, but its 1024-thread
// workgroups on every CU delayed the other generator's chain and the other batch: B=8 3.91 vs 3.86 ms per step with / without it.)
__global__ void __launch_bounds__(256) ff_compact_taps_kernel(const float *__restrict__ w, float *__restrict__ wc, size_t pairs) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // (co, ci) pair
    if (i >= pairs) return;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) wc[i * 3 + kd] = w[i * 27 + kd * 9 + 4];
}

static int ff_level(int Co, int D, int H, int W) {
    if (Co == 256 && D == 4 && H == 1 && W == 1) return 1;
    if (Co == 128 && D == 8 && H == 2 && W == 2) return 2;
    if (Co == 64 && D == 16 && H == 4 && W == 4) return 3;
    return 0;
}
static int ff_slices(int level) { return level == 3 ? 16 : 128; }

}  // namespace mphip

using namespace mphip;

extern "C" int mphip_flowfield_conv_gn_supported(int Ci, int Co, int D, int H, int W, int Cr, int groups) {
    const int lv = ff_level(Co, D, H, W);
    if (!lv || groups != 32 || Ci <= 0 || Ci % ff_slices(lv) || Cr < 0 || (Cr && Cr % ff_slices(lv))) return 0;
    return lv;
}

// Level 1 (4x1x1 volumes) can run from a compact copy of its weights: [Co][Ci][3], the taps (kd, 1, 1) — bytes, 0 for any other level
extern "C" size_t mphip_flowfield_compact_weight_bytes(int Ci, int Co, int D, int H, int W) {
    return (Ci > 0 && ff_level(Co, D, H, W) == 1) ? (size_t)Co * Ci * 3 * sizeof(float) : 0;
}
extern "C" int mphip_flowfield_compact_weight(const float *w, void *w_compact, int Ci, int Co, void *stream) {
    MPHIP_REQUIRE(w && w_compact && Ci > 0 && Co > 0, "flowfield_compact_weight: bad arguments");
    const size_t pairs = (size_t)Co * Ci;
    hipLaunchKernelGGL(ff_compact_taps_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (float *)w_compact, pairs);
    return check_launch("flowfield_compact_weight");
}

static int flowfield_conv_gn_impl(const float *x, const float *w, bool w_is_compact, const float *b, const float *gamma, const float *beta, const float *w2,
                                  const float *b2, const float *res_x, const float *res_w, const float *res_b, float *y, int N, int Ci,
                                  int Co, int D, int H, int W, int Cr, int uD, int uH, int uW, int groups, float eps, int relu,
                                  void *stream) {
    MPHIP_REQUIRE(x && w && gamma && beta && y, "flowfield_conv_gn: null pointer");
    MPHIP_REQUIRE((w2 == nullptr) == (b2 == nullptr), "flowfield_conv_gn: w2/b2 must both be set or both NULL");
    MPHIP_REQUIRE((res_x == nullptr) == (res_w == nullptr) && (res_x != nullptr) == (Cr > 0), "flowfield_conv_gn: residual input, weight and Cr go together");
    MPHIP_REQUIRE(N > 0 && uD >= 1 && uH >= 1 && uW >= 1, "flowfield_conv_gn: bad dims");
    const int lv = mphip_flowfield_conv_gn_supported(Ci, Co, D, H, W, Cr, groups);
    MPHIP_REQUIRE(lv, "flowfield_conv_gn: shape %d->%d @%dx%dx%d (Cr=%d, %d groups) is not a FlowField level (query mphip_flowfield_conv_gn_supported)",
                  Ci, Co, D, H, W, Cr, groups);
    FfParams p{x, w, b, gamma, beta, w2, b2, res_x, res_w, res_b, y, Ci, Co, Cr, uD, uH, uW, relu, eps};
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)(N * groups));
    switch (lv) {
        case 1:
            if (w_is_compact) hipLaunchKernelGGL((ff_block_plane_kernel<4, 1, 1, 8, 1024, true>), grid, dim3(1024), 0, s, p);
            else hipLaunchKernelGGL((ff_block_plane_kernel<4, 1, 1, 8, 1024>), grid, dim3(1024), 0, s, p);
            break;
        case 2: hipLaunchKernelGGL((ff_block_plane_kernel<8, 2, 2, 4, 512>), grid, dim3(512), 0, s, p); break;
        default: hipLaunchKernelGGL((ff_block_row_kernel<16, 4, 4, 2, 16>), grid, dim3(1024), 0, s, p); break;
    }
    return check_launch("flowfield_conv_gn");
}

extern "C" int mphip_flowfield_conv_gn(const float *x, const float *w, const float *b, const float *gamma, const float *beta, const float *w2,
                                       const float *b2, const float *res_x, const float *res_w, const float *res_b, float *y, int N, int Ci,
                                       int Co, int D, int H, int W, int Cr, int uD, int uH, int uW, int groups, float eps, int relu,
                                       void *stream) {
    return flowfield_conv_gn_impl(x, w, false, b, gamma, beta, w2, b2, res_x, res_w, res_b, y, N, Ci, Co, D, H, W, Cr, uD, uH, uW, groups, eps, relu, stream);
}
// the same with `w_compact` = the copy mphip_flowfield_compact_weight made of w (level 1 only: mphip_flowfield_compact_weight_bytes > 0); same bits
extern "C" int mphip_flowfield_conv_gn_compact(const float *x, const void *w_compact, const float *b, const float *gamma, const float *beta, const float *w2,
                                               const float *b2, const float *res_x, const float *res_w, const float *res_b, float *y, int N, int Ci,
                                               int Co, int D, int H, int W, int Cr, int uD, int uH, int uW, int groups, float eps, int relu,
                                               void *stream) {
    MPHIP_REQUIRE(mphip_flowfield_compact_weight_bytes(Ci, Co, D, H, W) > 0, "flowfield_conv_gn_compact: not a level with compact weights");
    return flowfield_conv_gn_impl(x, (const float *)w_compact, true, b, gamma, beta, w2, b2, res_x, res_w, res_b, y, N, Ci, Co, D, H, W, Cr, uD, uH, uW, groups, eps,
                                  relu, stream);
}

extern "C" size_t mphip_flowfield_out_workspace_bytes(int N) { return N > 0 ? (size_t)N * FO_G * 2 * sizeof(double) + (size_t)N * 3 * FO_S * sizeof(float) : 0; }

extern "C" int mphip_flowfield_out(const float *x, const float *w, const float *b, const float *gamma, const float *beta, float *em, int N,
                                   float eps, void *workspace, size_t workspace_bytes, void *stream) {
    MPHIP_REQUIRE(x && w && gamma && beta && em, "flowfield_out: null pointer");
    MPHIP_REQUIRE(N > 0, "flowfield_out: bad batch");
    const size_t need = mphip_flowfield_out_workspace_bytes(N);
    if (!workspace || workspace_bytes < need) {
        set_error("flowfield_out: workspace %zu bytes < required %zu", workspace_bytes, need);
        return MPHIP_EWORKSPACE;
    }
    double *part = (double *)workspace;
    float *y = (float *)((char *)workspace + (size_t)N * FO_G * 2 * sizeof(double));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ff_out_conv_kernel, dim3(N * FO_G), dim3(256), 0, s, x, w, b, y, part);
    hipLaunchKernelGGL(ff_out_norm_kernel, dim3(N * FO_G), dim3(256), 0, s, (const float *)y, (const double *)part, gamma, beta, eps, em);
    return check_launch("flowfield_out");
}

#ifdef MPHIP_FF_TRACE
extern "C" int mphip_debug_ff_trace(unsigned long long *host_out /* 512 * 8 */) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mphip::g_ff_trace), sizeof(unsigned long long) * 512 * 8) == hipSuccess ? 0 : -1;
}
#endif
