// K4/K5 — Conv3d 3x3x3 (pad 1) and 1x1x1 as implicit GEMM on the fp32 matrix cores.
// Reference call sites: nn.Conv3d at model.py:505,507,510,591 (G3d/ResBlock3D), 374-380,458
// (FlowField/ResBlock3D_Adaptive) and the 1x1 Conv2d at model.py:446.
//
// GEMM view (per conv):  Y[co][vox] = sum_{tap,ci} Wp[tap][ci][co] * X[ci][vox + tap]
//   M = Co (MFMA rows), N = voxels N*D*H*W (MFMA cols, contiguous in NCDHW so the C/D fragment's
//   32 lanes store 128 contiguous bytes), K = taps*Ci.
// MFMA: v_mfma_f32_32x32x2_f32 — exact fp32 (bitwise an fmaf chain), 64 FLOP/clk/SIMD.
//   A fragment = weights  A[i=lane&31][k=lane>>5] -> Wp[tap][ci+k][co0+i]   (one dword per lane)
//   B fragment = voxels   B[k=lane>>5][j=lane&31] -> X[ci+k][vox0+j + tap]  (one dword per lane)
//   C/D: col j = lane&31, row i = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "mphip_common.h"
#include "mphip_conv.h"

namespace mphip {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    return buf_load_f(rsrc, voff, soff);
}

// OIDHW [Co,Ci,k,k,k] -> [k^3][CiP][CoP], zero padded.
// transposed: `w` is the ORIGINAL conv's weight [Ci][Co][taps] and the packed conv is its bwd-data conv:
// Wt[co][ci][tap] = w[ci][co][taps-1-tap] (flipping all three axes reverses the linear tap index).
__device__ __forceinline__ void pack_weight_body(const float *__restrict__ w, float *__restrict__ wp, int Co, int Ci, int CoP,
                                                 int CiP, int taps, int transposed, unsigned bid, unsigned nblk) {
    size_t n = (size_t)taps * CiP * CoP;
    for (size_t i = (size_t)bid * blockDim.x + threadIdx.x; i < n; i += (size_t)nblk * blockDim.x) {
        int co = (int)(i % CoP);
        int ci = (int)((i / CoP) % CiP);
        int tap = (int)(i / ((size_t)CoP * CiP));
        const size_t src = transposed ? ((size_t)ci * Co + co) * taps + (taps - 1 - tap) : ((size_t)co * Ci + ci) * taps + tap;
        wp[i] = (co < Co && ci < Ci) ? w[src] : 0.0f;
    }
}
__global__ void pack_weight_kernel(const float *__restrict__ w, float *__restrict__ wp, int Co, int Ci, int CoP,
                                   int CiP, int taps, int transposed) {
    pack_weight_body(w, wp, Co, Ci, CoP, CiP, taps, transposed, blockIdx.x, gridDim.x);
}
// every exact-fp32 pack of a table in one launch (mphip_pack_table_run): a block finds its job among the selected ones
__global__ void __launch_bounds__(256) pack_many_f32_kernel(const PackJob *__restrict__ jobs, const int *__restrict__ sel,
                                                            const int *__restrict__ first, int n) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (first[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const PackJob j = jobs[sel[lo]];
    pack_weight_body(j.w, (float *)j.wp, j.Co, j.Ci, (j.Co + 31) / 32 * 32, (j.Ci + 1) / 2 * 2, j.k * j.k * j.k, j.transposed,
                     blockIdx.x - first[lo], first[lo + 1] - first[lo]);
}

// Generic gather variant: both operands straight from global/L2 (buffer loads, hardware zero fill
// for the padding halo), any D,H,W, batch folded into the voxel axis.  Block = 4 waves laid out
// WCO (along Co) x 4/WCO (along voxels); each wave owns MT x NT tiles of 32x32.
// gridDim = (voxel tiles, co tiles, split-K slices of the Ci range).
template <int KS, int MT, int NT, int WCO, bool SKIP>
__global__ void __launch_bounds__(256)
conv3d_gather_kernel(const float *__restrict__ x, const float *__restrict__ wp, const float *__restrict__ bias,
                     float *__restrict__ y, int N, int Ci, int CiP, int Co, int CoP, int D, int H, int W,
                     int ci_per_split, unsigned x_bytes) {
    MPHIP_LATENCY_KERNEL_PRIO();
    constexpr int TAPS = KS * KS * KS;
    constexpr int WVOX = 4 / WCO;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave_co = wave % WCO, wave_vox = wave / WCO;
    const int j = lane & 31, kk = lane >> 5;
    const int HW = H * W, DHW = D * HW;
    const long M = (long)N * DHW;

    const int co0 = (blockIdx.y * WCO + wave_co) * MT * 32;
    const long v0 = ((long)blockIdx.x * WVOX + wave_vox) * NT * 32;
    if (co0 >= CoP || v0 >= M) return;  // wave-uniform

    const int ci_begin = blockIdx.z * ci_per_split;
    const int ci_end = min(CiP, ci_begin + ci_per_split);

    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)x_bytes, 0x00020000);

    // per-lane voxel bookkeeping for each of the wave's NT column tiles
    unsigned vbyte[NT];   // byte offset of (n, ci=kk, d, h, w) in x
    unsigned vmask[NT];   // bit tap -> that neighbour exists (inside the volume)
    unsigned anymask = 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        long vg = v0 + t * 32 + j;
        bool ok = vg < M;
        long vv = ok ? vg : 0;
        int n = (int)(vv / DHW);
        int r = (int)(vv - (long)n * DHW);
        int d = r / HW, h = (r / W) % H, w = r % W;
        vbyte[t] = (unsigned)(((long)n * Ci * DHW + (long)kk * DHW + r) * 4);
        unsigned m = 0;
        if (ok) {
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                int kd = KS == 3 ? tap / 9 - 1 : 0, kh = KS == 3 ? (tap / 3) % 3 - 1 : 0, kw = KS == 3 ? tap % 3 - 1 : 0;
                bool in = (unsigned)(d + kd) < (unsigned)D && (unsigned)(h + kh) < (unsigned)H &&
                          (unsigned)(w + kw) < (unsigned)W;
                m |= in ? (1u << tap) : 0u;
            }
        }
        vmask[t] = m;
        anymask |= m;
    }
    if (SKIP) {  // wave-wide OR: taps no lane needs are skipped (tiny volumes: most taps are padding)
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) anymask |= __shfl_xor(anymask, s, 64);
        anymask = __builtin_amdgcn_readfirstlane(anymask);
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.0f;

    const float *wlane = wp + (size_t)kk * CoP + co0 + j;

    // All fragment loads of a channel pair are issued before its first MFMA, and (when the register budget
    // allows: MT+NT <= 3) the next pair's loads are issued before this pair's MFMAs.  The small-volume
    // layers this kernel serves (FlowField, 1x1 shortcuts) run about one wave per SIMD on cold, read-once
    // weights: a load -> wait -> MFMA chain per tap would pay a full memory latency 27 times per pair.
    // "Items" of one pipeline stage: the 27 taps of a channel pair (k=3), or 8 consecutive channel pairs (k=1).
    constexpr int ITEMS = KS == 3 ? TAPS : 8;
    constexpr int CI_STEP = KS == 3 ? 2 : 16;
#define MPHIP_GATHER_LOAD(A_, B_, ci_)                                                                               \
    {                                                                                                                \
        _Pragma("unroll") for (int it = 0; it < ITEMS; ++it) {                                                       \
            const int tap = KS == 3 ? it : 0;                                                                        \
            const int cc_ = (ci_) + (KS == 3 ? 0 : 2 * it);                                                          \
            if (SKIP && !((anymask >> tap) & 1u)) continue;                                                          \
            const bool in_ = cc_ < ci_end;               /* k=1: the last stage of a slice may be partial */         \
            const bool ci_ok_ = in_ && (cc_ + kk < Ci);  /* CiP may exceed Ci by one (odd Ci) */                     \
            const unsigned soff_ = (unsigned)((long)(in_ ? cc_ : ci_begin) * DHW * 4);                               \
            const int kd = KS == 3 ? tap / 9 - 1 : 0, kh = KS == 3 ? (tap / 3) % 3 - 1 : 0, kw = KS == 3 ? tap % 3 - 1 : 0; \
            const int toff = (kd * HW + kh * W + kw) * 4;                                                            \
            const float *wrow = wlane + ((size_t)tap * CiP + (in_ ? cc_ : ci_begin)) * CoP;                          \
            /* out-of-slice items read a valid (clamped) weight row: their B operand is the hardware zero fill, */   \
            /* so the product vanishes without a branch around the load                                         */   \
            _Pragma("unroll") for (int m = 0; m < MT; ++m) A_[it][m] = wrow[m * 32];                                 \
            _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                                         \
                bool ok = ((vmask[t] >> tap) & 1u) && ci_ok_;                                                        \
                B_[it][t] = buf_load(rsrc, ok ? vbyte[t] + (unsigned)toff : OOB, soff_);                             \
            }                                                                                                        \
        }                                                                                                            \
    }
#define MPHIP_GATHER_MFMA(A_, B_)                                                                                    \
    {                                                                                                                \
        _Pragma("unroll") for (int it = 0; it < ITEMS; ++it) {                                                       \
            if (SKIP && !((anymask >> (KS == 3 ? it : 0)) & 1u)) continue;                                           \
            _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                           \
                _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                       \
                    acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(A_[it][m], B_[it][t], acc[m][t], 0, 0, 0);      \
        }                                                                                                            \
    }
    constexpr bool DOUBLE_BUF = KS == 1 || (MT + NT) <= 3;
    if (DOUBLE_BUF) {
        float a0[ITEMS][MT], b0[ITEMS][NT], a1[ITEMS][MT], b1[ITEMS][NT];
        if (ci_begin < ci_end) MPHIP_GATHER_LOAD(a0, b0, ci_begin);
        for (int ci = ci_begin; ci < ci_end; ci += 2 * CI_STEP) {
            if (ci + CI_STEP < ci_end) MPHIP_GATHER_LOAD(a1, b1, ci + CI_STEP);
            MPHIP_GATHER_MFMA(a0, b0);
            if (ci + 2 * CI_STEP < ci_end) MPHIP_GATHER_LOAD(a0, b0, ci + 2 * CI_STEP);
            if (ci + CI_STEP < ci_end) MPHIP_GATHER_MFMA(a1, b1);
        }
    } else {
        float a[ITEMS][MT], b[ITEMS][NT];
        for (int ci = ci_begin; ci < ci_end; ci += CI_STEP) {
            MPHIP_GATHER_LOAD(a, b, ci);
            MPHIP_GATHER_MFMA(a, b);
        }
    }
#undef MPHIP_GATHER_LOAD
#undef MPHIP_GATHER_MFMA

    // epilogue: gridDim.z == 1 -> y (+bias); else partial slab z (bias added by the reduce kernel)
    const bool direct = gridDim.z == 1;
    float *dst = direct ? y : y + (size_t)blockIdx.z * N * Co * DHW;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        long vg = v0 + t * 32 + j;
        if (vg >= M) continue;
        int n = (int)(vg / DHW);
        int r = (int)(vg - (long)n * DHW);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                int co = co0 + m * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kk;
                if (co < Co) {
                    float vout = acc[m][t][reg];
                    if (direct && bias) vout += bias[co];
                    dst[((size_t)n * Co + co) * DHW + r] = vout;
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// LDS-tiled variant for the G3d levels (volume dims multiples of the tile, Co multiple of MT*32).
// A workgroup (4 waves) owns an output tile of TD x TH x TW voxels x (MT*32) channels.  Per chunk
// of KC input channels it stages (a) the input halo tile [KC][TD+2][TH+2][TW+2] (zero filled by
// the buffer-load range check) and (b) the weight slab [27][KC][MT*32] into LDS, double buffered
// through registers (loads for chunk c+1 are issued before the MFMAs of chunk c, written to the
// other buffer after them; one barrier per chunk).  The inner loop is then 27 taps x (MT+NT)
// ds_read_b32 with immediate offsets + MT*NT MFMAs — no address arithmetic, no global latency.
template <int TD, int TH, int TW, int MT, int KC>
__global__ void __launch_bounds__(256)
conv3d_k3_tiled_kernel(const float *__restrict__ x, const float *__restrict__ wp, const float *__restrict__ bias,
                       float *__restrict__ y, int N, int Ci, int CiP, int Co, int CoP, int D, int H, int W,
                       int chunks_per_split, unsigned x_bytes) {
    constexpr int TVOX = TD * TH * TW;
    constexpr int NT = TVOX / 128;  // 32-voxel column tiles per wave
    static_assert(TVOX % 128 == 0 && KC % 2 == 0, "tile shape");
    constexpr int HD = TD + 2, HH = TH + 2, HWp = TW + 2;
    constexpr int XS_PLANE = HD * HH * HWp;
    constexpr int CO_T = MT * 32;
    constexpr int XS_BUF = KC * XS_PLANE;       // floats per X buffer
    constexpr int WS_BUF = 27 * KC * CO_T;      // floats per W buffer
    constexpr int XE = (XS_BUF + 255) / 256;    // X elements staged per thread
    constexpr int WE = (WS_BUF / 4 + 255) / 256;  // W float4 staged per thread
    // buffers are padded to a whole number of per-thread pieces so staging needs no predicates
    constexpr int XS_PAD = XE * 256, WS_PAD = WE * 256 * 4;
    __shared__ __attribute__((aligned(16))) float smem[2 * XS_PAD + 2 * WS_PAD];
    float *const Ws = smem;                 // [2][27][KC][CO_T]   (first: 16-byte aligned float4 stores)
    float *const Xs = smem + 2 * WS_PAD;    // [2][KC][HD][HH][HWp]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kk = lane >> 5;
    const int HW = H * W, DHW = D * HW;

    // block -> (n, tile origin), co tile, chunk range
    const int tiles_w = W / TW, tiles_h = H / TH, tiles_d = D / TD;
    int bid = blockIdx.x;
    const int tw = bid % tiles_w; bid /= tiles_w;
    const int th = bid % tiles_h; bid /= tiles_h;
    const int td = bid % tiles_d;
    const int n = bid / tiles_d;
    const int d0 = td * TD, h0 = th * TH, w0 = tw * TW;
    const int co0 = blockIdx.y * CO_T;
    const int nchunks = Ci / KC;
    const int c_begin = blockIdx.z * chunks_per_split;
    const int c_end = min(nchunks, c_begin + chunks_per_split);

    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)x_bytes, 0x00020000);

    // staging plan (fixed per thread for the whole K loop)
    unsigned xsrc[XE];  // byte offset in x of this thread's halo elements for ci = 0 (OOB if padding)
#pragma unroll
    for (int i = 0; i < XE; ++i) {
        const int e = i * 256 + tid;
        unsigned off = OOB;
        if (e < XS_BUF) {
            const int kc = e / XS_PLANE, r = e % XS_PLANE;
            const int gd = d0 - 1 + r / (HH * HWp), gh = h0 - 1 + (r / HWp) % HH, gw = w0 - 1 + r % HWp;
            if ((unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W)
                off = (unsigned)((((long)n * Ci + kc) * DHW + (long)gd * HW + gh * W + gw) * 4);
        }
        xsrc[i] = off;
    }
    const float *wsrc[WE];  // this thread's float4 pieces of the weight slab for ci = 0
#pragma unroll
    for (int i = 0; i < WE; ++i) {
        const int f = i * 256 + tid;
        const int L = (f < WS_BUF / 4 ? f : 0) * 4;
        const int co = L % CO_T, kc = (L / CO_T) % KC, tap = L / (CO_T * KC);
        wsrc[i] = wp + ((size_t)tap * CiP + kc) * CoP + co0 + co;
    }

    float xr[XE];
#define MPHIP_ISSUE_LOADS(chunk)                                                                          \
    {                                                                                                     \
        const int ci0_ = (chunk) * KC;                                                                    \
        const unsigned soff_ = (unsigned)((long)ci0_ * DHW * 4);                                          \
        _Pragma("unroll") for (int i = 0; i < XE; ++i) xr[i] = buf_load(rsrc, xsrc[i], soff_);            \
    }
    /* weight slab: LDS-DMA, 16 B per lane, wave-uniform destination + lane*16 (lane-linear image) */     \
#define MPHIP_DMA_W(chunk, buf)                                                                           \
    {                                                                                                     \
        const size_t woff_ = (size_t)(chunk) * KC * CoP;                                                  \
        _Pragma("unroll") for (int i = 0; i < WE; ++i) __builtin_amdgcn_global_load_lds(                  \
            (const __attribute__((address_space(1))) void *)(wsrc[i] + woff_),                            \
            (__attribute__((address_space(3))) void *)(Ws + (buf) * WS_PAD + (i * 256 + wave * 64) * 4), 16, 0, 0); \
    }
#define MPHIP_WRITE_LDS(buf)                                                                              \
    {                                                                                                     \
        _Pragma("unroll") for (int i = 0; i < XE; ++i) Xs[(buf) * XS_PAD + i * 256 + tid] = xr[i];        \
    }

    // fragment read bases (floats): A = Ws[tap][kc=kk][m*32 + j], B = Xs[kc=kk][voxel + tap]
    const int a_base = kk * CO_T + j;
    int b_base[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int v = (wave * NT + t) * 32 + j;
        const int vw = v % TW, vh = (v / TW) % TH, vd = v / (TW * TH);
        b_base[t] = kk * XS_PLANE + (vd * HH + vh) * HWp + vw;
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.0f;

    if (c_begin < c_end) {
        MPHIP_DMA_W(c_begin, 0);
        MPHIP_ISSUE_LOADS(c_begin);
        MPHIP_WRITE_LDS(0);
        __syncthreads();
    }
    for (int c = c_begin; c < c_end; ++c) {
        const int buf = (c - c_begin) & 1;
        const bool more = c + 1 < c_end;
        if (more) {
            MPHIP_DMA_W(c + 1, buf ^ 1);
            MPHIP_ISSUE_LOADS(c + 1);
        }
        const float *wsb = Ws + buf * WS_PAD + a_base;
        const float *xsb = Xs + buf * XS_PAD;
#pragma unroll
        for (int kp = 0; kp < KC / 2; ++kp) {
#pragma unroll
            for (int tap = 0; tap < 27; ++tap) {
                constexpr int dummy = 0;
                (void)dummy;
                const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
                const int toff = (kd * HH + kh) * HWp + kw + kp * 2 * XS_PLANE;
                float a[MT], b[NT];
#pragma unroll
                for (int m = 0; m < MT; ++m) a[m] = wsb[(tap * KC + kp * 2) * CO_T + m * 32];
#pragma unroll
                for (int t = 0; t < NT; ++t) b[t] = xsb[b_base[t] + toff];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[t], acc[m][t], 0, 0, 0);
            }
        }
        if (more) MPHIP_WRITE_LDS(buf ^ 1);
        __syncthreads();
    }

#undef MPHIP_ISSUE_LOADS
#undef MPHIP_DMA_W
#undef MPHIP_WRITE_LDS
    const bool direct = gridDim.z == 1;
    float *dst = direct ? y : y + (size_t)blockIdx.z * N * Co * DHW;
    // bias for this lane's MT*16 output rows, fetched up front (one batch of loads, not one per store)
    float bv[MT][16];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
            bv[m][reg] = (direct && bias) ? bias[co0 + m * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kk] : 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int v = (wave * NT + t) * 32 + j;
        const int vw = v % TW, vh = (v / TW) % TH, vd = v / (TW * TH);
        float *dv = dst + (size_t)n * Co * DHW + (size_t)(d0 + vd) * HW + (h0 + vh) * W + w0 + vw;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int co = co0 + m * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kk;
                dv[(size_t)co * DHW] = acc[m][t][reg] + bv[m][reg];
            }
        }
    }
}

// y = bias + sum_z partial[z]  (z ascending: deterministic)
__global__ void __launch_bounds__(256)
splitk_reduce_kernel(const float *__restrict__ partial, const float *__restrict__ bias, float *__restrict__ y,
                     size_t n_out, int Co, int DHW, int splits) {
    MPHIP_LATENCY_KERNEL_PRIO();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    float s = sum_slabs(partial, splits, n_out, i);
    if (bias) s += bias[(i / DHW) % Co];
    y[i] = s;
}

struct ConvPlan {
    int MT, NT, WCO, splits, ci_per_split, CoP, CiP;
    bool skip;
    int tiled;  // 0 = gather kernel, 4/2 = LDS-tiled kernel with a (tiled,8,8) voxel tile
    dim3 grid;
};

constexpr int TILED_KC = 2;

static ConvPlan plan_conv(int N, int Ci, int Co, int D, int H, int W, int k) {
    ConvPlan p;
    p.CoP = (Co + 31) / 32 * 32;
    p.CiP = (Ci + 1) / 2 * 2;
    const long M = (long)N * D * H * W;
    const int co_tiles32 = p.CoP / 32;
    p.NT = (M >= 64 * 64 && k == 3) ? 2 : 1;  // k=1: short K loop, favour more (lighter) waves: 2 per SIMD
    const long vox_tiles = (M + p.NT * 32 - 1) / (p.NT * 32);
    // Largest split-K factor the channel range allows (>= 8 channels per slice, power of two, <= 32).
    static const long target = getenv("MPHIP_GATHER_TARGET_WAVES") ? atol(getenv("MPHIP_GATHER_TARGET_WAVES")) : 1024;   // dev: sweep
    static const int min_ch = getenv("MPHIP_GATHER_MIN_CH") ? atoi(getenv("MPHIP_GATHER_MIN_CH")) : 8;
    static const int split_cap = getenv("MPHIP_GATHER_MAX_SPLITS") ? atoi(getenv("MPHIP_GATHER_MAX_SPLITS")) : 32;
    int max_splits = 1;
    while (max_splits < split_cap && p.CiP / (max_splits * 2) >= min_ch && p.CiP % (max_splits * 4) == 0) max_splits *= 2;
    // Rows of 32 output channels per wave: as many as possible (operand reuse) while the launch still
    // offers ~one wave per SIMD (1024); small volumes trade reuse for parallelism (MT -> 1, more slices),
    // because there a wave's serial MFMA chain, not bandwidth, is the critical path.
    const int cands[4] = {4, 3, 2, 1};
    p.MT = 1;
    for (int ci = 0; ci < 4; ++ci) {
        const int mt = cands[ci];
        if (co_tiles32 % mt) continue;
        if ((long)(co_tiles32 / mt) * vox_tiles * max_splits >= target || mt == 1) {
            p.MT = mt;
            break;
        }
    }
    const int co_wave_tiles = co_tiles32 / p.MT;
    // waves along Co when the voxel axis is too short to feed 4 waves
    p.WCO = 1;
    if (vox_tiles < 4 * 64 && co_wave_tiles % 4 == 0) p.WCO = 4;
    else if (vox_tiles < 4 * 64 && co_wave_tiles % 2 == 0) p.WCO = 2;
    const int wvox = 4 / p.WCO;
    p.grid.x = (unsigned)((vox_tiles + wvox - 1) / wvox);
    p.grid.y = (unsigned)((co_wave_tiles + p.WCO - 1) / p.WCO);
    int splits = 1;
    while ((long)co_wave_tiles * vox_tiles * splits < target && splits < max_splits) splits *= 2;
    p.splits = splits;
    p.ci_per_split = p.CiP / splits;
    p.grid.z = splits;
    p.skip = (k == 3) && (D < 3 || H < 3 || W < 3);
    p.tiled = 0;
    if (k == 3 && Co % 96 == 0 && Ci % TILED_KC == 0 && H % 8 == 0 && W % 8 == 0 && D % 2 == 0 && !getenv("MPHIP_CONV_GATHER")) {
        p.tiled = D % 4 == 0 ? 4 : 2;
        const long tiles = (long)N * (D / p.tiled) * (H / 8) * (W / 8);
        const int nchunks = Ci / TILED_KC;
        p.grid = dim3((unsigned)tiles, Co / 96, 1);
        int sp = 1;
        while (tiles * (Co / 96) * sp < 512 && nchunks / (sp * 2) >= 12) sp *= 2;
        p.splits = sp;
        p.ci_per_split = (nchunks + sp - 1) / sp;  // chunks per split for the tiled kernel
        p.grid.z = sp;
    }
    return p;
}

}  // namespace mphip

using namespace mphip;

static size_t packed_elems_f32(int Co, int Ci, int k) {
    return (size_t)k * k * k * ((Ci + 1) / 2 * 2) * ((Co + 31) / 32 * 32);
}

extern "C" int mphip_conv3d_supported(int N, int Ci, int Co, int D, int H, int W, int k, int precision) {
    if (N <= 0 || Ci <= 0 || Co <= 0 || D <= 0 || H <= 0 || W <= 0 || (k != 1 && k != 3)) return 0;
    if ((size_t)N * Ci * D * H * W * 4 >= 0x80000000ull) return 0;
    if (precision == 0) return 1;
    if (precision == 1) return f16x3_supported(N, Ci, Co, D, H, W, k) ? 1 : 0;
    return 0;
}

extern "C" size_t mphip_packed_weight_bytes(int Co, int Ci, int k, int precision) {
    if (Co <= 0 || Ci <= 0 || (k != 1 && k != 3)) return 0;
    if (precision == 0) return packed_elems_f32(Co, Ci, k) * sizeof(float);
    if (precision == 1 && k == 3 && Ci % 16 == 0 && Co % 96 == 0) return f16x3_packed_bytes(Co, Ci);
    if (precision == 1 && k == 1 && Ci % 16 == 0 && Co % 96 == 0) return f16x3_packed_bytes_k1(Co, Ci);
    return 0;
}

static int pack_conv_weight(const float *w, void *wp, int Co, int Ci, int k, int precision, int transposed, const void *header_from,
                            void *stream);

extern "C" int mphip_pack_conv_weight(const float *w, void *wp, int Co, int Ci, int k, int precision, void *stream) {
    return pack_conv_weight(w, wp, Co, Ci, k, precision, 0, nullptr, stream);
}

// packs the bwd-data conv of the conv whose OIDHW weight is `w` [Ci][Co][k^3] (Co/Ci are the bwd-data conv's own
// output/input channels = the original conv's input/output channels): no flipped/transposed copy is materialised
extern "C" int mphip_pack_conv_weight_bwd_data(const float *w, void *wp, int Co, int Ci, int k, int precision, void *stream) {
    return pack_conv_weight(w, wp, Co, Ci, k, precision, 1, nullptr, stream);
}

// same, reusing the per-tensor scale header of an f16x3 pack of the SAME weight tensor (its forward pack): training
// re-packs both directions every step, and max|w| does not depend on the direction — saves the second absmax pass
extern "C" int mphip_pack_conv_weight_bwd_data_like(const float *w, void *wp, int Co, int Ci, int k, int precision,
                                                    const void *fwd_pack, void *stream) {
    MPHIP_REQUIRE(precision != 1 || fwd_pack, "pack_conv_weight_bwd_data_like: null forward pack");
    return pack_conv_weight(w, wp, Co, Ci, k, precision, 1, precision == 1 ? fwd_pack : nullptr, stream);
}

static int pack_conv_weight(const float *w, void *wp, int Co, int Ci, int k, int precision, int transposed, const void *header_from,
                            void *stream) {
    MPHIP_REQUIRE(w && wp, "pack_conv_weight: null pointer");
    MPHIP_REQUIRE(Co > 0 && Ci > 0 && (k == 1 || k == 3), "pack_conv_weight: bad dims");
    MPHIP_REQUIRE(mphip_packed_weight_bytes(Co, Ci, k, precision) > 0,
                  "pack_conv_weight: precision %d not available for Co=%d Ci=%d k=%d", precision, Co, Ci, k);
    if (precision == 1) return f16x3_pack(w, wp, Co, Ci, k, transposed, header_from, (hipStream_t)stream);
    size_t n = packed_elems_f32(Co, Ci, k);
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (float *)wp, Co, Ci,
                       (Co + 31) / 32 * 32, (Ci + 1) / 2 * 2, k * k * k, transposed);
    return check_launch("pack_conv_weight");
}

// ---- batched re-packing: a table of (weight, pack) jobs resolved once, run in <= 5 launches (zero headers, absmax, f16x3 k=3, f16x3
// k=1, exact fp32), capturable in a hipGraph.  Same bits as one mphip_pack_conv_weight* call per job.
namespace {
struct PackTable {
    int n = 0;
    void *dev = nullptr;          // [jobs | int arrays]
    const PackJob *jobs = nullptr;
    PackSel sel[4] = {};          // absmax, f16x3 k=3, f16x3 k=1, fp32
};
}  // namespace

extern "C" int mphip_pack_table_create(const mphip_pack_job *jobs, int n, void **table_out) {
    MPHIP_REQUIRE(jobs && table_out && n > 0, "pack_table_create: null pointer / no jobs");
    std::vector<PackJob> dj((size_t)n);
    std::vector<int> selv[4], firstv[4];
    for (int i = 0; i < n; ++i) {
        const mphip_pack_job &u = jobs[i];
        MPHIP_REQUIRE(u.w && u.wp, "pack_table_create: job %d: null pointer", i);
        MPHIP_REQUIRE(u.Co > 0 && u.Ci > 0 && (u.k == 1 || u.k == 3), "pack_table_create: job %d: bad dims", i);
        MPHIP_REQUIRE(mphip_packed_weight_bytes(u.Co, u.Ci, u.k, u.precision) > 0,
                      "pack_table_create: job %d: precision %d not available for Co=%d Ci=%d k=%d", i, u.precision, u.Co, u.Ci, u.k);
        PackJob &j = dj[(size_t)i];
        j.w = u.w; j.wp = u.wp; j.like = u.precision == 1 ? u.like : nullptr;
        j.Co = u.Co; j.Ci = u.Ci; j.k = u.k; j.precision = u.precision; j.transposed = u.transposed ? 1 : 0; j.reserved = 0;
        j.wino_off = (u.precision == 1 && u.k == 3) ? f16x3_pack_wino_offset(u.Co, u.Ci) : 0;
        auto add = [&](int kind, int blocks) {   // (firstv holds block COUNTS here; the running sums are formed below)
            selv[kind].push_back(i);
            firstv[kind].push_back(blocks);
        };
        if (u.precision == 1) {
            if (!j.like) add(0, f16x3_pack_blocks(j, 0));
            add(u.k == 3 ? 1 : 2, f16x3_pack_blocks(j, u.k == 3 ? 1 : 2));
        } else {
            const size_t ne = packed_elems_f32(u.Co, u.Ci, u.k);
            add(3, (int)std::min<size_t>(4096, (ne + 255) / 256));
        }
    }
    PackTable *t = new PackTable();
    t->n = n;
    size_t ints = 0;
    for (int k = 0; k < 4; ++k) ints += 2 * selv[k].size() + 1;
    const size_t jobs_bytes = ((size_t)n * sizeof(PackJob) + 15) / 16 * 16;
    std::vector<char> host(jobs_bytes + ints * sizeof(int));
    memcpy(host.data(), dj.data(), (size_t)n * sizeof(PackJob));
    if (hipMalloc(&t->dev, host.size()) != hipSuccess) {
        delete t;
        set_error("pack_table_create: hipMalloc of %zu bytes failed", host.size());
        return MPHIP_ELAUNCH;
    }
    t->jobs = (const PackJob *)t->dev;
    int *hi = (int *)(host.data() + jobs_bytes);
    const int *di = (const int *)((const char *)t->dev + jobs_bytes);
    size_t at = 0;
    for (int k = 0; k < 4; ++k) {
        const int m = (int)selv[k].size();
        t->sel[k].n = m;
        t->sel[k].job = di + at;
        for (int q = 0; q < m; ++q) hi[at + q] = selv[k][q];
        at += (size_t)m;
        t->sel[k].first = di + at;
        int run = 0;
        for (int q = 0; q < m; ++q) { hi[at + q] = run; run += firstv[k][q]; }
        hi[at + m] = run;
        t->sel[k].blocks = run;
        at += (size_t)m + 1;
    }
    if (hipMemcpy(t->dev, host.data(), host.size(), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(t->dev);
        delete t;
        set_error("pack_table_create: upload failed");
        return MPHIP_ELAUNCH;
    }
    *table_out = t;
    return MPHIP_OK;
}

extern "C" int mphip_pack_table_run(void *table, void *stream) {
    MPHIP_REQUIRE(table, "pack_table_run: null table");
    const PackTable *t = (const PackTable *)table;
    hipStream_t s = (hipStream_t)stream;
    // (a `like` header is written by the absmax launch of THIS run when its owner is a job of the table, or earlier by the caller:
    //  either way before the pack launches below, which are the first to read it)
    if (t->sel[0].n || t->sel[1].n || t->sel[2].n) {
        const int rc = f16x3_pack_many(t->jobs, t->sel[0], t->sel[1], t->sel[2], s);
        if (rc != MPHIP_OK) return rc;
    }
    if (t->sel[3].n)
        hipLaunchKernelGGL(pack_many_f32_kernel, dim3((unsigned)t->sel[3].blocks), dim3(256), 0, s, t->jobs, t->sel[3].job, t->sel[3].first,
                           t->sel[3].n);
    return check_launch("pack_table_run");
}

extern "C" int mphip_pack_table_destroy(void *table) {
    if (!table) return MPHIP_OK;
    PackTable *t = (PackTable *)table;
    (void)hipFree(t->dev);
    delete t;
    return MPHIP_OK;
}

constexpr size_t RANGE_BYTES = ((MPHIP_RANGE_FLOATS * sizeof(float) + 255) / 256) * 256;  // descriptor + padding: keeps what follows aligned

static size_t conv_ws_bytes(int N, int Ci, int Co, int D, int H, int W, int k, int precision, bool roi);
// GroupNorm statistics from the f16x3 3x3x3 kernel's epilogue (unsplit launches): bytes of its per-(tile, wave, channel) partials, 0 = the
// separate statistics pass is used
static size_t gn_epilogue_bytes(int N, int Ci, int Co, int D, int H, int W, int k, int precision, F16x3Plan *plan_out = nullptr) {
    static const bool off = getenv("MPHIP_GN_EPILOGUE") && getenv("MPHIP_GN_EPILOGUE")[0] == '0';   // dev: same-box A/B
    if (off || precision != 1 || k != 3 || !mphip_conv3d_supported(N, Ci, Co, D, H, W, k, precision)) return 0;
    const F16x3Plan fp = f16x3_plan(N, Ci, Co, D, H, W, false);
    if (fp.splits != 1) return 0;
    if (fp.variant == 4 && D == 2) return 0;   // (the F(2,3) kernel's two-frame mode leaves no partials: separate statistics pass)
    if (plan_out) *plan_out = fp;
    return (size_t)fp.grid.x * f16x3_tile_waves(fp) * Co * 2 * sizeof(float) + 256;   // (+ the accumulator unscale behind the partials)
}
static size_t conv_gn_bytes(int N, int Ci, int Co, int D, int H, int W, int k, int precision, int gn_groups) {
    const size_t a = groupnorm_ws_bytes(N, Co, D * H * W, gn_groups), b = gn_epilogue_bytes(N, Ci, Co, D, H, W, k, precision);
    return a > b ? a : b;
}
extern "C" size_t mphip_conv3d_workspace_bytes(int N, int Ci, int Co, int D, int H, int W, int k, int precision) {
    return conv_ws_bytes(N, Ci, Co, D, H, W, k, precision, false);
}
static size_t conv_ws_bytes(int N, int Ci, int Co, int D, int H, int W, int k, int precision, bool roi) {
    if (!mphip_conv3d_supported(N, Ci, Co, D, H, W, k, precision)) return 0;
    int splits = precision == 1 ? (k == 1 ? 1 : f16x3_plan(N, Ci, Co, D, H, W, roi).splits) : plan_conv(N, Ci, Co, D, H, W, k).splits;
    // precision 1: room in front for the range descriptor the library computes itself when the caller passes none
    return (precision == 1 ? RANGE_BYTES : 0) + (splits > 1 ? (size_t)splits * N * Co * D * H * W * sizeof(float) : 0);
}

template <int KS, int MT, int NT, int WCO, bool SKIP>
static void launch_gather(const ConvPlan &p, const float *x, const float *wp, const float *bias, float *dst, int N,
                          int Ci, int Co, int D, int H, int W, unsigned x_bytes, hipStream_t s) {
    hipLaunchKernelGGL((conv3d_gather_kernel<KS, MT, NT, WCO, SKIP>), p.grid, dim3(256), 0, s, x, wp, bias, dst, N, Ci,
                       p.CiP, Co, p.CoP, D, H, W, p.ci_per_split, x_bytes);
}

template <int KS, int MT, int NT, int WCO>
static void dispatch_skip(const ConvPlan &p, const float *x, const float *wp, const float *bias, float *dst, int N,
                          int Ci, int Co, int D, int H, int W, unsigned xb, hipStream_t s) {
    if (KS == 3 && p.skip) launch_gather<KS, MT, NT, WCO, true>(p, x, wp, bias, dst, N, Ci, Co, D, H, W, xb, s);
    else launch_gather<KS, MT, NT, WCO, false>(p, x, wp, bias, dst, N, Ci, Co, D, H, W, xb, s);
}

template <int KS, int MT, int NT>
static void dispatch_wco(const ConvPlan &p, const float *x, const float *wp, const float *bias, float *dst, int N,
                         int Ci, int Co, int D, int H, int W, unsigned xb, hipStream_t s) {
    switch (p.WCO) {
        case 4: dispatch_skip<KS, MT, NT, 4>(p, x, wp, bias, dst, N, Ci, Co, D, H, W, xb, s); break;
        case 2: dispatch_skip<KS, MT, NT, 2>(p, x, wp, bias, dst, N, Ci, Co, D, H, W, xb, s); break;
        default: dispatch_skip<KS, MT, NT, 1>(p, x, wp, bias, dst, N, Ci, Co, D, H, W, xb, s); break;
    }
}

template <int KS, int MT>
static void dispatch_nt(const ConvPlan &p, const float *x, const float *wp, const float *bias, float *dst, int N,
                        int Ci, int Co, int D, int H, int W, unsigned xb, hipStream_t s) {
    if (p.NT == 2) dispatch_wco<KS, MT, 2>(p, x, wp, bias, dst, N, Ci, Co, D, H, W, xb, s);
    else dispatch_wco<KS, MT, 1>(p, x, wp, bias, dst, N, Ci, Co, D, H, W, xb, s);
}

template <int KS>
static void dispatch_mt(const ConvPlan &p, const float *x, const float *wp, const float *bias, float *dst, int N,
                        int Ci, int Co, int D, int H, int W, unsigned xb, hipStream_t s) {
    switch (p.MT) {
        case 4: dispatch_nt<KS, 4>(p, x, wp, bias, dst, N, Ci, Co, D, H, W, xb, s); break;
        case 3: dispatch_nt<KS, 3>(p, x, wp, bias, dst, N, Ci, Co, D, H, W, xb, s); break;
        case 2: dispatch_nt<KS, 2>(p, x, wp, bias, dst, N, Ci, Co, D, H, W, xb, s); break;
        default: dispatch_nt<KS, 1>(p, x, wp, bias, dst, N, Ci, Co, D, H, W, xb, s); break;
    }
}

// Measurement hook (bench.py through the plan): the NEXT conv launch on this thread is bracketed by the two HIP events, recorded on
// the launch stream right before and right after the conv kernel itself (before the split-K reduce / GroupNorm statistics).
static thread_local hipEvent_t g_time_e0 = nullptr, g_time_e1 = nullptr;
extern "C" void mphip_conv3d_time_next_launch(void *e0, void *e1) {
    g_time_e0 = (hipEvent_t)e0;
    g_time_e1 = (hipEvent_t)e1;
}

static int conv3d_run(const float *x, const float *in_affine, int in_relu, const float *x_range, const void *w_packed, const float *bias, float *y,
                      float *gn_stats, int gn_groups, float gn_eps, bool keep_split, int N, int Ci, int Co, int D, int H, int W, int k, int precision, void *workspace,
                      size_t workspace_bytes, void *stream, const int *roi = nullptr, int roi_frames = 0, int roi_dilate = 0,
                      const GnTable *gn_table = nullptr) {
    // the measurement hook's events belong to THIS call whatever happens below (an early error return must not leave them armed
    // for an unrelated later launch on this thread)
    hipEvent_t te0 = g_time_e0, te1 = g_time_e1;
    g_time_e0 = g_time_e1 = nullptr;
    MPHIP_REQUIRE(x && w_packed && y, "conv3d_fwd: null pointer");
    MPHIP_REQUIRE(N > 0 && Ci > 0 && Co > 0 && D > 0 && H > 0 && W > 0, "conv3d_fwd: bad dims");
    MPHIP_REQUIRE(k == 1 || k == 3, "conv3d_fwd: kernel size %d not supported (1 or 3)", k);
    MPHIP_REQUIRE(precision == 0 || precision == 1, "conv3d_fwd: precision %d not supported", precision);
    const size_t x_bytes = (size_t)N * Ci * D * H * W * sizeof(float);
    MPHIP_REQUIRE(x_bytes < 0x80000000ull, "conv3d_fwd: input of %zu bytes exceeds the 2 GiB buffer-addressing limit",
                  x_bytes);
    MPHIP_REQUIRE(mphip_conv3d_supported(N, Ci, Co, D, H, W, k, precision),
                  "conv3d_fwd: precision %d not available for this shape (query mphip_conv3d_supported)", precision);
    MPHIP_REQUIRE(!in_affine || precision == 1, "conv3d_fwd: the fused input GroupNorm needs the f16x3 kernel (precision 1)");
    hipStream_t s = (hipStream_t)stream;
    ConvPlan p{};
    F16x3Plan fp{};
    int splits;
    if (precision == 1 && k == 1) {
        MPHIP_REQUIRE(!in_affine, "conv3d_fwd: the fused input GroupNorm is for the 3x3x3 f16x3 kernel");
        splits = 1;   // the k=1 GEMM kernel (conv3d_f16x3.hip): no split-K
    } else if (precision == 1) {
        fp = f16x3_plan(N, Ci, Co, D, H, W, roi != nullptr);
        splits = fp.splits;
    } else {
        p = plan_conv(N, Ci, Co, D, H, W, k);
        splits = p.splits;
    }
    float *dst = y;
    const size_t n_out = (size_t)N * Co * D * H * W;
    // f16x3: the input's range descriptor (mphip_common.h).  None given -> one extra pass computes max|x| into the head
    // of the workspace (a fused input GroupNorm changes the values: its descriptor must come from
    // mphip_groupnorm_affine_table).
    const size_t range_bytes = precision == 1 ? RANGE_BYTES : 0;
    if (precision == 1 && !x_range) {
        MPHIP_REQUIRE(!in_affine, "conv3d_fwd: the fused input GroupNorm needs the range descriptor of mphip_groupnorm_affine_table");
        if (!workspace || workspace_bytes < range_bytes) {
            set_error("conv3d_fwd: workspace %zu bytes < required %zu (no x_range given: the input's range descriptor goes there)", workspace_bytes, range_bytes);
            return MPHIP_EWORKSPACE;
        }
        int rc0 = absmax_range_launch(x, (size_t)N * Ci * D * H * W, (float *)workspace, s);
        if (rc0) return rc0;
        x_range = (const float *)workspace;
    }
    workspace = workspace ? (char *)workspace + range_bytes : nullptr;
    workspace_bytes = workspace_bytes > range_bytes ? workspace_bytes - range_bytes : 0;
    const size_t slab_bytes = (splits > 1 && !keep_split) ? (size_t)splits * n_out * sizeof(float) : 0;
    const size_t gn_bytes = gn_stats ? conv_gn_bytes(N, Ci, Co, D, H, W, k, precision, gn_groups) : 0;
    const bool gn_in_epilogue = gn_stats && !roi && !keep_split && gn_epilogue_bytes(N, Ci, Co, D, H, W, k, precision) > 0;
    if (slab_bytes + gn_bytes > 0) {
        if (!workspace || workspace_bytes < slab_bytes + gn_bytes) {
            set_error("conv3d_fwd: workspace %zu bytes < required %zu", workspace_bytes, slab_bytes + gn_bytes);
            return MPHIP_EWORKSPACE;
        }
    }
    if (keep_split) {
        dst = y;  // caller-owned [splits][N,Co,D,H,W]; the kernels add the bias only when splits == 1
    } else if (splits > 1) {
        dst = (float *)workspace;
    }
    void *gn_ws = (char *)workspace + slab_bytes;
    int rc;
    const bool stamp = te0 && te1 && precision == 1 && k == 3;   // the f16x3 3x3x3 kernels carry the events themselves (kernel begin / end)
    if (te0 && !stamp) (void)hipEventRecord(te0, s);
    if (precision == 1 && k == 1) {
        rc = f16x3_launch_k1(x, w_packed, bias, dst, N, Ci, Co, D * H * W, x_range, s);
    } else if (precision == 1) {
        int *tile_list = nullptr;
        if (roi) {   // the tile list lives at the END of the caller's workspace (mphip_conv3d_roi_workspace_bytes)
            const size_t list_bytes = (((size_t)fp.grid.x + 1) * sizeof(int) + 255) / 256 * 256;
            if (workspace_bytes < (slab_bytes + gn_bytes + 255) / 256 * 256 + list_bytes) {
                set_error("conv3d_fwd_roi: workspace too small for the tile list (query mphip_conv3d_roi_workspace_bytes)");
                return MPHIP_EWORKSPACE;
            }
            // at a fixed 256-byte-aligned offset behind the slabs / statistics partials (not "the end of whatever the caller passed": a
            // larger workspace whose size is not a multiple of 4 would misalign the int list — ADVICE r3)
            tile_list = (int *)((char *)workspace + (slab_bytes + gn_bytes + 255) / 256 * 256);
        }
        rc = f16x3_launch(fp, x, w_packed, bias, dst, N, Ci, Co, D, H, W, in_affine, in_relu, x_range, s, roi, roi_frames, tile_list, roi_dilate,
                          gn_in_epilogue ? (float *)gn_ws : nullptr, stamp ? te0 : nullptr, stamp ? te1 : nullptr);
    } else {
        const float *wf = (const float *)w_packed;
        if (p.tiled == 4)
            hipLaunchKernelGGL((conv3d_k3_tiled_kernel<4, 8, 8, 3, TILED_KC>), p.grid, dim3(256), 0, s, x, wf, bias, dst, N, Ci,
                               p.CiP, Co, p.CoP, D, H, W, p.ci_per_split, (unsigned)x_bytes);
        else if (p.tiled == 2)
            hipLaunchKernelGGL((conv3d_k3_tiled_kernel<2, 8, 8, 3, TILED_KC>), p.grid, dim3(256), 0, s, x, wf, bias, dst, N, Ci,
                               p.CiP, Co, p.CoP, D, H, W, p.ci_per_split, (unsigned)x_bytes);
        else if (k == 3) dispatch_mt<3>(p, x, wf, bias, dst, N, Ci, Co, D, H, W, (unsigned)x_bytes, s);
        else dispatch_mt<1>(p, x, wf, bias, dst, N, Ci, Co, D, H, W, (unsigned)x_bytes, s);
        rc = check_launch("conv3d_fwd");
    }
    if (te1 && !stamp) (void)hipEventRecord(te1, s);
    if (rc) return rc;
    const int S = D * H * W;
    if (splits > 1 && !keep_split) {
        // (a reduce fused with the GroupNorm statistics was tried: one workgroup per (sample, group) span is too
        //  little parallelism for these small tensors — 4 % slower end to end than reduce + single-launch stats)
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(n_out, 256)), dim3(256), 0, s, (const float *)workspace, bias, y,
                           n_out, Co, S, splits);
        rc = check_launch("conv3d_fwd(splitk_reduce)");
        if (rc) return rc;
    }
    if (gn_stats && gn_in_epilogue)
        rc = groupnorm_stats_from_tiles((const float *)gn_ws, bias, gn_stats, N, Co, S, gn_groups, gn_eps, (int)fp.grid.x / N, f16x3_tile_waves(fp), s, gn_table);
    else if (gn_stats)
        rc = groupnorm_stats_launch(y, gn_stats, N, Co, S, gn_groups, gn_eps, gn_ws, s, gn_table);
    return rc;
}

extern "C" int mphip_conv3d_fwd(const float *x, const float *x_range, const void *w_packed, const float *bias, float *y, int N,
                                int Ci, int Co, int D, int H, int W, int k, int precision, void *workspace,
                                size_t workspace_bytes, void *stream) {
    return conv3d_run(x, nullptr, 0, x_range, w_packed, bias, y, nullptr, 0, 0.0f, false, N, Ci, Co, D, H, W, k, precision, workspace,
                      workspace_bytes, stream);
}

// Demand-driven conv (include/mphip.h): only the output tiles that intersect the consumer's sample boxes are computed.
extern "C" int mphip_conv3d_roi_granule(int N, int Ci, int Co, int D, int H, int W, int k, int precision, int *tile_dhw) {
    if (!tile_dhw || precision != 1 || k != 3 || !mphip_conv3d_supported(N, Ci, Co, D, H, W, k, 1)) return 0;
    f16x3_tile_dims(f16x3_plan(N, Ci, Co, D, H, W, true), tile_dhw);
    return 1;
}

extern "C" size_t mphip_conv3d_roi_workspace_bytes(int N, int Ci, int Co, int D, int H, int W, int k, int precision) {
    int dims[3];
    if (!mphip_conv3d_roi_granule(N, Ci, Co, D, H, W, k, precision, dims)) return mphip_conv3d_workspace_bytes(N, Ci, Co, D, H, W, k, precision);
    const size_t base = conv_ws_bytes(N, Ci, Co, D, H, W, k, precision, true);
    const size_t tiles = (size_t)N * (D / dims[0]) * (H / dims[1]) * (W / dims[2]);
    return (base + 255) / 256 * 256 + ((tiles + 1) * sizeof(int) + 255) / 256 * 256;
}

extern "C" int mphip_conv3d_fwd_roi(const float *x, const float *x_range, const void *w_packed, const float *bias, float *y, const int *roi,
                                    int roi_frames, int N, int Ci, int Co, int D, int H, int W, int k, int precision, void *workspace,
                                    size_t workspace_bytes, void *stream) {
    MPHIP_REQUIRE(roi, "conv3d_fwd_roi: null box list (mphip_warp_sample_box makes it)");
    MPHIP_REQUIRE(roi_frames >= 0 && (roi_frames == 0 || N == 1), "conv3d_fwd_roi: roi_frames > 0 (several boxes on one volume) needs N == 1");
    int dims[3];
    const bool tiled = mphip_conv3d_roi_granule(N, Ci, Co, D, H, W, k, precision, dims) != 0;
    // (shapes / precisions without a tiled kernel compute every voxel: the result is a superset of what was asked for)
    return conv3d_run(x, nullptr, 0, x_range, w_packed, bias, y, nullptr, 0, 0.0f, false, N, Ci, Co, D, H, W, k, precision, workspace,
                      workspace_bytes, stream, tiled ? roi : nullptr, tiled ? roi_frames : 0);
}

// which kernel a (full) conv launch of this shape takes: 0 = exact fp32 kernels / k = 1 GEMM, 1 / 2 = the direct f16x3 kernel on
// (td,8,8) / (4,8,16) tiles, 5 = the 1-D Winograd F(2,3) f16x3 kernel.  Measurement / tests only.
extern "C" int mphip_conv3d_kernel_variant(int N, int Ci, int Co, int D, int H, int W, int k, int precision) {
    if (precision != 1 || k != 3 || !mphip_conv3d_supported(N, Ci, Co, D, H, W, k, precision)) return 0;
    return f16x3_plan(N, Ci, Co, D, H, W).variant + 1;
}

extern "C" int mphip_conv3d_splits(int N, int Ci, int Co, int D, int H, int W, int k, int precision) {
    if (!mphip_conv3d_supported(N, Ci, Co, D, H, W, k, precision)) return 0;
    if (precision == 1 && k == 1) return 1;
    return precision == 1 ? f16x3_plan(N, Ci, Co, D, H, W).splits : plan_conv(N, Ci, Co, D, H, W, k).splits;
}

extern "C" int mphip_conv3d_fwd_split(const float *x, const float *x_range, const void *w_packed, const float *bias, float *out,
                                      int N, int Ci, int Co, int D, int H, int W, int k, int precision, void *workspace,
                                      size_t workspace_bytes, void *stream) {
    // workspace: only a library-computed range descriptor (precision 1 with x_range == NULL)
    return conv3d_run(x, nullptr, 0, x_range, w_packed, bias, out, nullptr, 0, 0.0f, true, N, Ci, Co, D, H, W, k, precision, workspace,
                      workspace_bytes, stream);
}

extern "C" size_t mphip_conv3d_gn_workspace_bytes(int N, int Ci, int Co, int D, int H, int W, int k, int precision,
                                                  int gn_groups) {
    if (gn_groups <= 0 || Co % gn_groups) return 0;
    return mphip_conv3d_workspace_bytes(N, Ci, Co, D, H, W, k, precision) + conv_gn_bytes(N, Ci, Co, D, H, W, k, precision, gn_groups);
}

extern "C" int mphip_conv3d_gn_fwd(const float *x, const float *x_range, const void *w_packed, const float *bias, float *y, float *gn_stats,
                                   int N, int Ci, int Co, int D, int H, int W, int k, int precision, int gn_groups,
                                   float gn_eps, void *workspace, size_t workspace_bytes, void *stream) {
    MPHIP_REQUIRE(gn_stats, "conv3d_gn_fwd: null stats pointer");
    MPHIP_REQUIRE(gn_groups > 0 && Co > 0 && Co % gn_groups == 0, "conv3d_gn_fwd: Co=%d not divisible into %d groups", Co,
                  gn_groups);
    return conv3d_run(x, nullptr, 0, x_range, w_packed, bias, y, gn_stats, gn_groups, gn_eps, false, N, Ci, Co, D, H, W, k, precision, workspace,
                      workspace_bytes, stream);
}

// conv + the statistics of the GroupNorm that follows + that norm's affine table and range bound (what mphip_groupnorm_affine_table
// writes), all in the one call: the norm is then folded into the NEXT conv with mphip_conv3d_gnin_*_fwd(table, table_range).
extern "C" int mphip_conv3d_gn_table_fwd(const float *x, const float *x_range, const void *w_packed, const float *bias, float *y, float *gn_stats,
                                         const float *gamma, const float *beta, const float *w2, const float *b2, float *table,
                                         float *table_range, int N, int Ci, int Co, int D, int H, int W, int k, int precision, int gn_groups,
                                         float gn_eps, void *workspace, size_t workspace_bytes, void *stream) {
    MPHIP_REQUIRE(gn_stats && gamma && beta && table && table_range, "conv3d_gn_table_fwd: null pointer");
    MPHIP_REQUIRE((w2 == nullptr) == (b2 == nullptr), "conv3d_gn_table_fwd: w2/b2 must both be set or both NULL");
    MPHIP_REQUIRE(gn_groups > 0 && Co > 0 && Co % gn_groups == 0, "conv3d_gn_table_fwd: Co=%d not divisible into %d groups", Co, gn_groups);
    GnTable t;
    t.gamma = gamma; t.beta = beta; t.w2 = w2; t.b2 = b2; t.table = table; t.range = table_range;
    return conv3d_run(x, nullptr, 0, x_range, w_packed, bias, y, gn_stats, gn_groups, gn_eps, false, N, Ci, Co, D, H, W, k, precision, workspace,
                      workspace_bytes, stream, nullptr, 0, 0, &t);
}

extern "C" int mphip_conv3d_gnin_fwd(const float *x, const float *in_affine, const float *x_range, int in_relu, const void *w_packed,
                                     const float *bias, float *y, int N, int Ci, int Co, int D, int H, int W, int k,
                                     int precision, void *workspace, size_t workspace_bytes, void *stream) {
    MPHIP_REQUIRE(in_affine && x_range, "conv3d_gnin_fwd: null affine table / range descriptor (mphip_groupnorm_affine_table makes both)");
    return conv3d_run(x, in_affine, in_relu, x_range, w_packed, bias, y, nullptr, 0, 0.0f, false, N, Ci, Co, D, H, W, k, precision,
                      workspace, workspace_bytes, stream);
}

// conv(relu?(GroupNorm(x))) with the statistics of its own output for the NEXT GroupNorm (conv2 of a residual block)
extern "C" int mphip_conv3d_gnin_gn_fwd(const float *x, const float *in_affine, const float *x_range, int in_relu, const void *w_packed,
                                        const float *bias, float *y, float *gn_stats, int N, int Ci, int Co, int D, int H,
                                        int W, int k, int precision, int gn_groups, float gn_eps, void *workspace,
                                        size_t workspace_bytes, void *stream) {
    MPHIP_REQUIRE(in_affine && x_range && gn_stats, "conv3d_gnin_gn_fwd: null pointer");
    MPHIP_REQUIRE(gn_groups > 0 && Co > 0 && Co % gn_groups == 0, "conv3d_gnin_gn_fwd: Co=%d not divisible into %d groups", Co,
                  gn_groups);
    return conv3d_run(x, in_affine, in_relu, x_range, w_packed, bias, y, gn_stats, gn_groups, gn_eps, false, N, Ci, Co, D, H, W, k,
                      precision, workspace, workspace_bytes, stream);
}

// bwd-data of a conv whose output gradient dy is zero outside per-frame boxes (the gradient of a gather, mphip_warp_volume_bwd):
// dx is zero outside the boxes grown by one voxel — dx is zero-filled and only the tiles those grown boxes touch are computed.
extern "C" int mphip_conv3d_bwd_data_roi(const float *dy, const void *wt_packed, float *dx, const float *dy_scale, const int *roi, int N,
                                         int Ci, int Co, int D, int H, int W, int k, int precision, void *workspace,
                                         size_t workspace_bytes, void *stream) {
    MPHIP_REQUIRE(roi && dx, "conv3d_bwd_data_roi: null pointer");
    MPHIP_REQUIRE(precision == 0 || dy_scale, "conv3d_bwd_data_roi: the f16x3 kernel needs the gradient scale of mphip_grad_prep");
    int dims[3];
    // (a split-K launch reduces its slabs over EVERY voxel — garbage outside the listed tiles, where dx must be zero: such
    //  shapes, small volumes only, take the full evaluation)
    const bool tiled = mphip_conv3d_roi_granule(N, Ci, Co, D, H, W, k, precision, dims) != 0 && f16x3_plan(N, Ci, Co, D, H, W, true).splits == 1;
    if (tiled) zero_fill(dx, (size_t)N * Co * D * H * W * sizeof(float), (hipStream_t)stream);   // (16-byte multiple: W % 8 == 0 on this path)
    return conv3d_run(dy, nullptr, 0, precision == 1 ? dy_scale : nullptr, wt_packed, nullptr, dx, nullptr, 0, 0.0f, false, N, Ci, Co, D, H, W, k,
                      precision, workspace, workspace_bytes, stream, tiled ? roi : nullptr, 0, 1);
}

// bwd-data of a conv = the forward conv of dy with the flipped / transposed weight (packed by the caller from
// Wt[ci][co][a][b][c] = W[co][ci][k-1-a][k-1-b][k-1-c]); dy_scale (from mphip_grad_prep) gives the f16x3 kernel the
// gradient's own power-of-two scale instead of the activations' fixed one.  Ci/Co here are dy's / dx's channels.
extern "C" int mphip_conv3d_bwd_data(const float *dy, const void *wt_packed, float *dx, const float *dy_scale, int N, int Ci,
                                     int Co, int D, int H, int W, int k, int precision, void *workspace,
                                     size_t workspace_bytes, void *stream) {
    MPHIP_REQUIRE(precision == 0 || dy_scale, "conv3d_bwd_data: the f16x3 kernel needs the gradient scale of mphip_grad_prep");
    return conv3d_run(dy, nullptr, 0, precision == 1 ? dy_scale : nullptr, wt_packed, nullptr, dx, nullptr, 0, 0.0f, false, N, Ci,
                      Co, D, H, W, k, precision, workspace, workspace_bytes, stream);
}
