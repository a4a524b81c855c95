// conv3d 3x3x3 backward-weight on the f16 matrix cores with split precision ("f16x3", as the forward kernel):
//   dW[co][ci][tap] = sum_{n,v} dY[n][co][v] * X[n][ci][v + tap]
// Both operands are split v*S = hi + lo (two f16) while the tile is staged; each product is
// hi*hi + hi*lo + lo*hi accumulated in fp32 by v_mfma_f32_32x32x16_f16.  X uses the activations' fixed scale
// (16, as the forward pass); dY is a gradient of arbitrary magnitude and gets a per-tensor power-of-two scale
// (mphip_grad_prep: one pass over dY that also yields the bias gradient).
//
// GEMM view per tap: M = co, N = ci, K = voxels.  The MFMA wants 8 consecutive K values per lane: a lane's
// K group is one 8-voxel W row of the tile, so dY fragments are 16-B LDS reads, and the three kw taps of an X
// halo row (10 voxels) come from ONE 20-B read shifted with v_alignbit — no per-tap re-staging.
//
// Workgroup = 8 waves, one (96 co x 32 ci) block of all 27 taps, streaming 2x8x8-voxel tiles of its voxel range.
// 27 taps = 9 (kd,kh) rows: wave w owns row w (3 kw x 3 co-tiles = 9 accumulators); row 8 is spread over waves
// 0..2 (one co-tile each) so the four SIMDs carry 21/21/21/18 MFMA units instead of 27/18/18/18.
// Partial sums go to slab[blockIdx.z]; the ordered slab reduce makes the result deterministic.
// Staging goes through registers between two barriers (40 % of the kernel's time, not overlapped with the MFMAs: the
// accumulators leave no registers for a prefetch).  An asynchronous variant was built and measured — next tile's raw fp32
// rows by LDS-DMA during the MFMAs, then an LDS->LDS split pass, 1x8x8 tiles so both copies fit in 138 KB — and was 35 %
// SLOWER (0.82 vs 0.60 ms on the full-resolution layer): the conversion pass with its 2-byte LDS writes and the doubled
// barrier / halo count of the smaller tile cost more than the hidden latency.
#include "mphip_common.h"
#include "mphip_conv.h"

namespace mphip {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr float BF_CLAMP = 65000.0f;
constexpr int BF_AP = 136;                    // halves per co row of the dY tile: 128 voxels + 8 (pitch 68 dwords: b128 conflict-free)
constexpr int BF_XROW = 12;                   // halves per halo row: 10 voxels + 2 (24 B, 8-B aligned)
constexpr int BF_XCI = 4 * 10 * BF_XROW + 8;  // halves per input channel: 4 planes x 10 rows (+8: pitch 244 dwords)
constexpr int BF_A_PART = 96 * BF_AP;
constexpr int BF_X_PART = 32 * BF_XCI;

// (same split as conv3d_f16x3.hip: operands are pre-scaled by their tensors' range descriptors; a non-finite value is not
//  clamped but propagates — an overflowed gradient must stay visible to GradScaler, train.py:145,318-320)
__device__ __forceinline__ void bf_split(float v, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16)v;
    const float r = v - (float)hi;
    lo = (fabsf(v) <= BF_CLAMP) ? (_Float16)r : (_Float16)0.0f;
}

__device__ __forceinline__ half8 as_half8(unsigned a, unsigned b, unsigned c, unsigned d) {
    u32x4 u = {a, b, c, d};
    return __builtin_bit_cast(half8, u);
}

// the three kw-shifted 8-voxel fragments of one halo row (10 halves at `row`, 8-B aligned)
__device__ __forceinline__ void halo_row_frags(const _Float16 *row, half8 f[3]) {
    const uint2 a = *reinterpret_cast<const uint2 *>(row);
    const uint2 b = *reinterpret_cast<const uint2 *>(row + 4);
    const unsigned c = *reinterpret_cast<const unsigned *>(row + 8);
    f[0] = as_half8(a.x, a.y, b.x, b.y);
    f[1] = as_half8(__builtin_amdgcn_alignbit(a.y, a.x, 16), __builtin_amdgcn_alignbit(b.x, a.y, 16),
                    __builtin_amdgcn_alignbit(b.y, b.x, 16), __builtin_amdgcn_alignbit(c, b.y, 16));
    f[2] = as_half8(a.y, b.x, b.y, c);
}

#define BF_MFMA3(ACC, AH, AL, BH, BL)                                           \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL, BH, ACC, 0, 0, 0);         \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BL, ACC, 0, 0, 0);         \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BH, ACC, 0, 0, 0);

// SINGLE: the autocast(float16) policy (mphip_conv3d_set_half_products) — one product per multiply on the hi halves (= the
// operands rounded to f16, what ATen's autocast feeds its conv backward), fp32 accumulation: a third of the MFMAs, no lo planes staged.
template <bool SINGLE>
__global__ void __launch_bounds__(512)
conv_bwd_weight_f16x3_kernel(const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ gscale,
                             const float *__restrict__ x_range,
                             float *__restrict__ slabs, int N, int Ci, int Co, int D, int H, int W, int tiles_per_split,
                             const int *__restrict__ dy_boxes) {
    __shared__ __attribute__((aligned(16))) _Float16 smem[2 * BF_A_PART + 2 * BF_X_PART];
    _Float16 *const As = smem;                  // [part][co 96][BF_AP]
    _Float16 *const Xs = smem + 2 * BF_A_PART;  // [part][ci 32][plane 4][row 10][BF_XROW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kb = lane >> 5;
    const int HW = H * W, DHW = D * HW;
    const int ci_tiles = (Ci + 31) / 32;
    const int ci0 = (blockIdx.x % ci_tiles) * 32;
    const int co0 = (blockIdx.x / ci_tiles) * 96;
    const int tiles_w = W / 8, tiles_h = (H + 7) / 8, tiles_d = (D + 1) / 2;
    const int ntiles = N * tiles_d * tiles_h * tiles_w;
    const int t_begin = blockIdx.z * tiles_per_split;
    const int t_end = min(ntiles, t_begin + tiles_per_split);
    const float dscale = gscale[0];
    float BF_X_SCALE, bf_x_unscale;  // the saved activation's own operand scale (its range descriptor)
    range_scale_block(x_range, BF_X_SCALE, bf_x_unscale);
    const int kd = wave / 3, kh = wave % 3;  // this wave's tap row; waves 0..2 also take row 8 (kd=kh=2), co-tile `wave`
    const bool heavy = wave < 3;

    f32x16 acc[3][3], acc2[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[t][r] = 0.0f;
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][m][r] = 0.0f;
    }

    for (int tile = t_begin; tile < t_end; ++tile) {
        int r_ = tile;
        const int tw = r_ % tiles_w; r_ /= tiles_w;
        const int th = r_ % tiles_h; r_ /= tiles_h;
        const int td = r_ % tiles_d;
        const int n = r_ / tiles_d;
        const int d0 = td * 2, h0 = th * 8, w0 = tw * 8;
        if (dy_boxes) {   // dY is zero outside its frame's box {lx,ly,lz,ex,ey,ez} (the gradient of a gather): such tiles add nothing
            const int *b = dy_boxes + n * 8;
            if (!(w0 < b[0] + b[3] && w0 + 8 > b[0] && h0 < b[1] + b[4] && h0 + 8 > b[1] && d0 < b[2] + b[5] && d0 + 2 > b[2])) continue;
        }
        __syncthreads();  // the previous tile is fully consumed
        int tz = 0;
        asm volatile("" : "+v"(tz));  // opaque 0, new per tile: keeps the staging code's per-thread index math from being hoisted out of the
                                      // tile loop into registers the 192 accumulators do not leave (hipcc spilled 12 of them to scratch)
#ifdef BF_ABL_NOSTAGE
        if (tile == t_begin)
#endif
        {
        {   // ---- dY tile: 96 co x (2 d x 8 h) rows of 8 voxels, 3 rows per thread
            float4 ya[3][2];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int q = tid + i * 512 + tz, co = q >> 4, rr = q & 15;
                const int gd = d0 + (rr >> 3), gh = h0 + (rr & 7);
                const bool ok = co0 + co < Co && gd < D && gh < H;
                const float *p = dy + (ok ? ((size_t)n * Co + co0 + co) * DHW + (size_t)gd * HW + gh * W + w0 : 0);
                ya[i][0] = *reinterpret_cast<const float4 *>(p);
                ya[i][1] = *reinterpret_cast<const float4 *>(p + 4);
                if (!ok) ya[i][0] = ya[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int q = tid + i * 512 + tz, co = q >> 4, rr = q & 15;
                const float v[8] = {ya[i][0].x, ya[i][0].y, ya[i][0].z, ya[i][0].w, ya[i][1].x, ya[i][1].y, ya[i][1].z, ya[i][1].w};
                half8 hi, lo;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    _Float16 h, l;
                    bf_split(v[e] * dscale, h, l);
                    hi[e] = h;
                    lo[e] = l;
                }
                *reinterpret_cast<half8 *>(As + co * BF_AP + rr * 8) = hi;
                if constexpr (!SINGLE) *reinterpret_cast<half8 *>(As + BF_A_PART + co * BF_AP + rr * 8) = lo;
            }
        }
        {   // ---- X halo: 32 ci x 4 planes x 10 rows of 10 voxels (zero outside the volume), <= 3 rows per thread
            float4 xa[3][2];
            float xl[3], xr[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int q = tid + i * 512 + tz;
                const int ci = q / 40, pr = q % 40;
                const int gd = d0 - 1 + pr / 10, gh = h0 - 1 + pr % 10;
                const bool ok = q < 1280 && ci0 + ci < Ci && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H;
                const float *p = x + (ok ? ((size_t)n * Ci + ci0 + ci) * DHW + (size_t)gd * HW + gh * W + w0 : 0);
                const bool okl = ok && w0 > 0, okr = ok && w0 + 8 < W;
                xa[i][0] = *reinterpret_cast<const float4 *>(p);
                xa[i][1] = *reinterpret_cast<const float4 *>(p + 4);
                xl[i] = *(okl ? p - 1 : x);
                xr[i] = *(okr ? p + 8 : x);
                if (!ok) xa[i][0] = xa[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!okl) xl[i] = 0.0f;
                if (!okr) xr[i] = 0.0f;
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int q = tid + i * 512 + tz;
                if (q < 1280) {
                    const int ci = q / 40, pr = q % 40;
                    const float v[10] = {xl[i], xa[i][0].x, xa[i][0].y, xa[i][0].z, xa[i][0].w,
                                         xa[i][1].x, xa[i][1].y, xa[i][1].z, xa[i][1].w, xr[i]};
                    half2v hp[5], lp[5];
#pragma unroll
                    for (int e = 0; e < 5; ++e) {
                        _Float16 h0_, l0_, h1_, l1_;
                        bf_split(v[2 * e] * BF_X_SCALE, h0_, l0_);
                        bf_split(v[2 * e + 1] * BF_X_SCALE, h1_, l1_);
                        hp[e] = half2v{h0_, h1_};
                        lp[e] = half2v{l0_, l1_};
                    }
                    _Float16 *dst = Xs + ci * BF_XCI + pr * BF_XROW;
#pragma unroll
                    for (int e = 0; e < 5; ++e) {
                        *reinterpret_cast<half2v *>(dst + 2 * e) = hp[e];
                        if constexpr (!SINGLE) *reinterpret_cast<half2v *>(dst + BF_X_PART + 2 * e) = lp[e];
                    }
                }
            }
        }
        }
        __syncthreads();
#ifndef BF_ABL_NOMFMA
#pragma unroll 2
        for (int ks = 0; ks < 8; ++ks) {
            const int dl = ks >> 2, hrow = 2 * (ks & 3) + kb;
            half8 bh[3], bl[3];
            const _Float16 *xrow = Xs + j * BF_XCI + ((dl + kd) * 10 + hrow + kh) * BF_XROW;
            halo_row_frags(xrow, bh);
            if constexpr (!SINGLE) halo_row_frags(xrow + BF_X_PART, bl);
            const _Float16 *arow = As + j * BF_AP + ks * 16 + kb * 8;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const half8 ah = *reinterpret_cast<const half8 *>(arow + m * 32 * BF_AP);
                if constexpr (SINGLE) {
#pragma unroll
                    for (int t = 0; t < 3; ++t) acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[t], acc[t][m], 0, 0, 0);
                } else {
                    const half8 al = *reinterpret_cast<const half8 *>(arow + m * 32 * BF_AP + BF_A_PART);
#pragma unroll
                    for (int t = 0; t < 3; ++t) { BF_MFMA3(acc[t][m], ah, al, bh[t], bl[t]) }
                }
            }
            if (heavy) {
                half8 ch[3], cl[3];
                const _Float16 *xrow2 = Xs + j * BF_XCI + ((dl + 2) * 10 + hrow + 2) * BF_XROW;
                halo_row_frags(xrow2, ch);
                const half8 ah = *reinterpret_cast<const half8 *>(arow + wave * 32 * BF_AP);
                if constexpr (SINGLE) {
#pragma unroll
                    for (int t = 0; t < 3; ++t) acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ch[t], acc2[t], 0, 0, 0);
                } else {
                    halo_row_frags(xrow2 + BF_X_PART, cl);
                    const half8 al = *reinterpret_cast<const half8 *>(arow + wave * 32 * BF_AP + BF_A_PART);
#pragma unroll
                    for (int t = 0; t < 3; ++t) { BF_MFMA3(acc2[t], ah, al, ch[t], cl[t]) }
                }
            }
        }
#endif
    }

    // slab layout [27][Co][Ci] (a store covers 32 consecutive ci); C/D: column = lane&31 = ci, rows = co
    const float unscale = gscale[1] * bf_x_unscale;
    float *slab = slabs + (size_t)blockIdx.z * Co * Ci * 27;
    const int ci = ci0 + j;
    if (ci < Ci) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int tap = (kd * 3 + kh) * 3 + t;
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int co = co0 + m * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kb;
                    if (co < Co) slab[((size_t)tap * Co + co) * Ci + ci] = acc[t][m][reg] * unscale;
                }
            if (heavy) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int co = co0 + wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kb;
                    if (co < Co) slab[((size_t)(24 + t) * Co + co) * Ci + ci] = acc2[t][reg] * unscale;
                }
            }
        }
    }
}

// ---- 1x1x1 backward-weight: dW[co][ci] = sum_v dY[co][v] * X[ci][v] — a plain GEMM with K = voxels, HBM-bound (each
// operand is read once per output-tile column/row).  Workgroup = 8 waves on a 96 co x 96 ci output tile; a staged tile
// holds 128 voxels = 8 K-steps of 16, one per wave, so every wave keeps all nine 32x32 accumulators and the eight
// partial sums are combined in LDS before one slab write.  Same split precision as above.
__global__ void __launch_bounds__(512)
conv_bwd_weight_k1_f16x3_kernel(const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ gscale,
                                const float *__restrict__ x_range,
                                float *__restrict__ slabs, int N, int Ci, int Co, int DHW, int tiles_per_split) {
    // r05: the waves split the OUTPUT, not K.  Wave w owns the 32 x 32 tile (m, t) = (w / 3, w % 3) of the 96 x 96 block and walks all
    // eight K-steps of a staged tile itself; the ninth tile (2, 2) is split by K-step over the eight waves and is the only one that needs
    // a cross-wave sum.  (Until r04 every wave kept all nine accumulators for ONE K-step and the eight partial blocks were folded through
    // LDS in eight serial read-add-write rounds: ~10 us of a launch whose workgroups see two tiles each — 60 us for 38 MB of operands.)
    // 32 accumulator registers instead of 144 leave room to keep the next tile's global loads in flight under the MFMAs.
    __shared__ __attribute__((aligned(16))) _Float16 smem[4 * BF_A_PART];
    _Float16 *const As = smem;                  // dY [part][96][BF_AP]
    _Float16 *const Bs = smem + 2 * BF_A_PART;  // X  [part][96][BF_AP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kb = lane >> 5;
    const int ci_tiles = (Ci + 95) / 96;
    const int ci0 = (blockIdx.x % ci_tiles) * 96, co0 = (blockIdx.x / ci_tiles) * 96;
    const int tiles_per_sample = DHW / 128;
    const int ntiles = N * tiles_per_sample;
    const int t_begin = blockIdx.z * tiles_per_split, t_end = min(ntiles, t_begin + tiles_per_split);
    const float dscale = gscale[0];
    float BF_X_SCALE, bf_x_unscale;  // the saved activation's own operand scale (its range descriptor)
    range_scale_block(x_range, BF_X_SCALE, bf_x_unscale);
    const int om = wave / 3, ot = wave % 3;   // this wave's own output tile: co rows om*32.., ci columns ot*32..
    f32x16 acc, acc2;                         // own tile; this wave's K-step share of tile (2,2)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.0f; acc2[r] = 0.0f; }

    float4 va[2][3][2], vb[2][3][2];   // the staged tile's raw rows ([0: dY, 1: X][3 rows per thread][8 voxels]) and the next tile's, in flight
    auto prefetch = [&](int tile, float4 (&va)[2][3][2]) __attribute__((always_inline)) {
        const int n = tile / tiles_per_sample, v0 = (tile % tiles_per_sample) * 128;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const float *src = which ? x : dy;
            const int C = which ? Ci : Co, c0 = which ? ci0 : co0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int q = tid + i * 512, ch = q >> 4, grp = q & 15;
                const bool ok = c0 + ch < C;
                const float *p = src + (ok ? ((size_t)n * C + c0 + ch) * DHW + v0 + grp * 8 : 0);
                va[which][i][0] = *reinterpret_cast<const float4 *>(p);
                va[which][i][1] = *reinterpret_cast<const float4 *>(p + 4);
                if (!ok) va[which][i][0] = va[which][i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    if (t_begin < t_end) prefetch(t_begin, va);
    for (int tile = t_begin; tile < t_end; ++tile) {
        __syncthreads();   // the previous tile's fragments are read
#pragma unroll
        for (int which = 0; which < 2; ++which) {  // 0: dY -> As, 1: X -> Bs; 96 channels x 16 rows of 8 voxels, 3 rows per thread
            const float scale = which ? BF_X_SCALE : dscale;
            _Float16 *dst = which ? Bs : As;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int q = tid + i * 512, ch = q >> 4, grp = q & 15;
                const float4 a = va[which][i][0], b = va[which][i][1];
                const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                half8 hi, lo;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    _Float16 h, l;
                    bf_split(v[e] * scale, h, l);
                    hi[e] = h;
                    lo[e] = l;
                }
                *reinterpret_cast<half8 *>(dst + ch * BF_AP + grp * 8) = hi;
                *reinterpret_cast<half8 *>(dst + BF_A_PART + ch * BF_AP + grp * 8) = lo;
            }
        }
        __syncthreads();
        const bool more = tile + 1 < t_end;
        if (more) prefetch(tile + 1, vb);   // in flight under the MFMAs below
        const _Float16 *arow = As + (om * 32 + j) * BF_AP + kb * 8, *brow = Bs + (ot * 32 + j) * BF_AP + kb * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {   // the own tile: all eight K-steps of the staged tile, in order
            const half8 ah = *reinterpret_cast<const half8 *>(arow + ks * 16);
            const half8 al = *reinterpret_cast<const half8 *>(arow + ks * 16 + BF_A_PART);
            const half8 bh = *reinterpret_cast<const half8 *>(brow + ks * 16);
            const half8 bl = *reinterpret_cast<const half8 *>(brow + ks * 16 + BF_A_PART);
            BF_MFMA3(acc, ah, al, bh, bl)
        }
        {   // tile (2,2): K-step `wave`
            const int koff = wave * 16 + kb * 8;
            const half8 ah = *reinterpret_cast<const half8 *>(As + (64 + j) * BF_AP + koff);
            const half8 al = *reinterpret_cast<const half8 *>(As + (64 + j) * BF_AP + koff + BF_A_PART);
            const half8 bh = *reinterpret_cast<const half8 *>(Bs + (64 + j) * BF_AP + koff);
            const half8 bl = *reinterpret_cast<const half8 *>(Bs + (64 + j) * BF_AP + koff + BF_A_PART);
            BF_MFMA3(acc2, ah, al, bh, bl)
        }
        if (more) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) { va[a][b][0] = vb[a][b][0]; va[a][b][1] = vb[a][b][1]; }
        }
    }
    // the block's 96 x 96 sums in LDS: eight tiles straight from their owners, tile (2,2) as the ordered sum of the waves' shares
    __syncthreads();
    float *sum = reinterpret_cast<float *>(smem);       // [96][96]
    float *part = sum + 96 * 96;                         // [8 waves][16 regs][64 lanes]
    static_assert((96 * 96 + 8 * 16 * 64) * 4 <= 4 * BF_A_PART * 2, "LDS: combine region");
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        sum[(om * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kb) * 96 + ot * 32 + j] = acc[reg];
        part[(wave * 16 + reg) * 64 + lane] = acc2[reg];
    }
    __syncthreads();
    for (int i = tid; i < 16 * 64; i += 512) {   // (reg, lane) of tile (2,2): waves 0..7 in order
        const int reg = i >> 6, ln = i & 63;
        float a = part[i];
#pragma unroll
        for (int w = 1; w < 8; ++w) a += part[w * 1024 + i];
        sum[(64 + (reg & 3) + 8 * (reg >> 2) + 4 * (ln >> 5)) * 96 + 64 + (ln & 31)] = a;
    }
    __syncthreads();
    const float unscale = gscale[1] * bf_x_unscale;
    float *slab = slabs + (size_t)blockIdx.z * Co * Ci;
    for (int i = tid; i < 96 * 96; i += 512) {
        const int co = co0 + i / 96, ci = ci0 + i % 96;
        if (co < Co && ci < Ci) slab[(size_t)co * Ci + ci] = sum[i] * unscale;
    }
}

__global__ void __launch_bounds__(256)
slab_reduce_plain_kernel(const float *__restrict__ slabs, float *__restrict__ out, size_t n, int splits) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = sum_slabs(slabs, splits, n, i);
}

// ---- gradient preparation: per-(n,c) plane sums (-> bias gradient) and max|dy| (-> f16 scale) in one pass ----
constexpr int GP_CHUNK = 8192;  // floats per workgroup
__global__ void __launch_bounds__(256)
grad_prep_partial_kernel(const float *__restrict__ dy, float2 *__restrict__ partial, int S, int chunks) {
    const int plane = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    const float *p = dy + (size_t)plane * S;
    const int begin = chunk * GP_CHUNK, end = min(S, begin + GP_CHUNK);
    float s = 0.0f, m = 0.0f;
    if ((S & 3) == 0) {
        for (int i = begin + threadIdx.x * 4; i < end; i += 1024) {
            const float4 v = *reinterpret_cast<const float4 *>(p + i);
            s += (v.x + v.y) + (v.z + v.w);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    } else {
        for (int i = begin + threadIdx.x; i < end; i += 256) {
            const float v = p[i];
            s += v;
            m = fmaxf(m, fabsf(v));
        }
    }
    double acc = (double)s;  // <= 32 fp32 adds per thread, combined in double
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) {
        acc += __shfl_xor(acc, sft, 64);
        m = fmaxf(m, __shfl_xor(m, sft, 64));
    }
    __shared__ double red[4];
    __shared__ float redm[4];
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = acc;
        redm[threadIdx.x >> 6] = m;
    }
    __syncthreads();
    // (sum, max|.|) per workgroup; the finalize kernel folds them — no atomics (thousands of same-address atomicMax
    // from one launch serialise in L2 and cost more than the pass itself)
    if (threadIdx.x == 0)
        partial[blockIdx.x] = make_float2((float)((red[0] + red[1]) + (red[2] + red[3])),
                                          fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3])));
}

__global__ void __launch_bounds__(256)
grad_prep_finalize_kernel(const float2 *__restrict__ partial, float *__restrict__ scale, float *__restrict__ dbias, int N, int C,
                          int chunks) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0) {
        float m = 0.0f;
        for (int i = threadIdx.x; i < N * C * chunks; i += 256) m = fmaxf(m, partial[i].y);
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) m = fmaxf(m, __shfl_xor(m, sft, 64));
        __shared__ float redm[4];
        if ((threadIdx.x & 63) == 0) redm[threadIdx.x >> 6] = m;
        __syncthreads();
        m = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
        if (threadIdx.x == 0) {
        float sc = 1.0f;
        if (m > 0.0f && m < 1e30f) {
            int e;
            frexpf(m, &e);             // m < 2^e
            sc = ldexpf(1.0f, 14 - e);  // m * sc < 2^14
        }
        scale[0] = sc;
        scale[1] = 1.0f / sc;
        scale[2] = m;
        }
    }
    if (c < C && dbias) {
        double a = 0.0;
        for (int n = 0; n < N; ++n)
            for (int k = 0; k < chunks; ++k) a += (double)partial[((size_t)n * C + c) * chunks + k].x;
        dbias[c] = (float)a;
    }
}

// slabs [split][27][Co*Ci] (ci contiguous: the kernel's stores are coalesced) -> OIDHW [Co*Ci][27], summed in split
// order; 64 (co,ci) pairs per workgroup, transposed through LDS so reads and writes are both contiguous.
__global__ void __launch_bounds__(256)
slab_reduce_f16x3_kernel(const float *__restrict__ slabs, float *__restrict__ out, size_t ncc, int splits) {
    __shared__ float tile[64 * 27];
    const size_t cc0 = (size_t)blockIdx.x * 64;
    const int c = threadIdx.x & 63;
    for (int tap = threadIdx.x >> 6; tap < 27; tap += 4)
        if (cc0 + c < ncc) tile[c * 27 + tap] = sum_slabs(slabs, splits, ncc * 27, (size_t)tap * ncc + cc0 + c);
    __syncthreads();
    const size_t n_here = min((size_t)64, ncc - cc0) * 27;
    for (size_t i = threadIdx.x; i < n_here; i += 256) out[cc0 * 27 + i] = tile[i];
}

static void bwf_plan(int N, int Ci, int Co, int D, int H, int W, int &splits, int &tps) {
    const long ntiles = (long)N * ((D + 1) / 2) * ((H + 7) / 8) * (W / 8);
    const int bxy = ((Ci + 31) / 32) * ((Co + 95) / 96);
    long sp = bxy >= 256 ? 1 : 256 / bxy;           // one workgroup per CU (114 KB of LDS each): fill the chip once
    if (sp > ntiles) sp = ntiles;
    if (sp < 1) sp = 1;
    tps = (int)((ntiles + sp - 1) / sp);
    splits = (int)((ntiles + tps - 1) / tps);
}

static void bwf_k1_plan(int N, int Ci, int Co, int DHW, int &splits, int &tps) {
    const long ntiles = (long)N * (DHW / 128);
    const int bxy = ((Ci + 95) / 96) * ((Co + 95) / 96);
    long sp = bxy >= 256 ? 1 : 256 / bxy;
    if (sp > ntiles) sp = ntiles;
    if (sp < 1) sp = 1;
    tps = (int)((ntiles + sp - 1) / sp);
    splits = (int)((ntiles + tps - 1) / tps);
}

bool bwd_weight_f16x3_supported(int N, int Ci, int Co, int D, int H, int W, int k) {
    if (N <= 0 || Ci <= 0 || Co <= 0 || D <= 0 || H <= 0 || W <= 0) return false;
    if (k == 1) return ((long)D * H * W) % 128 == 0;
    return k == 3 && W % 8 == 0;
}

size_t bwd_weight_f16x3_ws_bytes(int N, int Ci, int Co, int D, int H, int W, int k) {
    int splits, tps;
    if (k == 1) {
        bwf_k1_plan(N, Ci, Co, D * H * W, splits, tps);
        return (size_t)splits * Co * Ci * sizeof(float);
    }
    bwf_plan(N, Ci, Co, D, H, W, splits, tps);
    return (size_t)splits * Co * Ci * 27 * sizeof(float);
}

int bwd_weight_f16x3_launch(const float *x, const float *x_range, const float *dy, const float *dy_scale, float *dw, int N, int Ci,
                            int Co, int D, int H, int W, int k, void *workspace, hipStream_t s, const int *dy_boxes) {
    int splits, tps;
    if (k == 1) {
        bwf_k1_plan(N, Ci, Co, D * H * W, splits, tps);
        dim3 grid(((Ci + 95) / 96) * ((Co + 95) / 96), 1, splits);
        hipLaunchKernelGGL(conv_bwd_weight_k1_f16x3_kernel, grid, dim3(512), 0, s, x, dy, dy_scale, x_range, (float *)workspace, N, Ci, Co,
                           D * H * W, tps);
        const size_t nw = (size_t)Co * Ci;
        hipLaunchKernelGGL(slab_reduce_plain_kernel, dim3(cdiv(nw, 256)), dim3(256), 0, s, (const float *)workspace, dw, nw, splits);
        return check_launch("conv3d_bwd_weight(f16x3, k=1)");
    }
    bwf_plan(N, Ci, Co, D, H, W, splits, tps);
    dim3 grid(((Ci + 31) / 32) * ((Co + 95) / 96), 1, splits);
    if (conv_half_products())   // the calling thread's autocast policy
        hipLaunchKernelGGL(conv_bwd_weight_f16x3_kernel<true>, grid, dim3(512), 0, s, x, dy, dy_scale, x_range, (float *)workspace, N, Ci, Co, D,
                           H, W, tps, dy_boxes);
    else
        hipLaunchKernelGGL(conv_bwd_weight_f16x3_kernel<false>, grid, dim3(512), 0, s, x, dy, dy_scale, x_range, (float *)workspace, N, Ci, Co, D,
                           H, W, tps, dy_boxes);
    const size_t ncc = (size_t)Co * Ci;
    hipLaunchKernelGGL(slab_reduce_f16x3_kernel, dim3(cdiv(ncc, 64)), dim3(256), 0, s, (const float *)workspace, dw, ncc, splits);
    return check_launch("conv3d_bwd_weight(f16x3)");
}

}  // namespace mphip

using namespace mphip;

extern "C" size_t mphip_grad_prep_workspace_bytes(int N, int C, int S) {
    return N > 0 && C > 0 && S > 0 ? (size_t)N * C * cdiv(S, GP_CHUNK) * sizeof(float2) : 0;
}

extern "C" int mphip_grad_prep(const float *dy, float *dbias, float *scale, int N, int C, int S, void *workspace,
                               size_t workspace_bytes, void *stream) {
    MPHIP_REQUIRE(dy && scale, "grad_prep: null pointer");
    MPHIP_REQUIRE(N > 0 && C > 0 && S > 0, "grad_prep: bad dims");
    const int chunks = cdiv(S, GP_CHUNK);
    if (!workspace || workspace_bytes < mphip_grad_prep_workspace_bytes(N, C, S)) {
        set_error("grad_prep: workspace %zu bytes < required %zu", workspace_bytes, mphip_grad_prep_workspace_bytes(N, C, S));
        return MPHIP_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(grad_prep_partial_kernel, dim3(N * C * chunks), dim3(256), 0, s, dy, (float2 *)workspace, S, chunks);
    hipLaunchKernelGGL(grad_prep_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, (const float2 *)workspace, scale, dbias, N,
                       C, chunks);
    return check_launch("grad_prep");
}
