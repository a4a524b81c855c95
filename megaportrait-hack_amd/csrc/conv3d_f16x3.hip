// K4 fast mode — Conv3d 3x3x3 on the f16 matrix cores with fp32-class accuracy ("f16x3" split).
//
// Why: the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the f16 MFMA rate and the
// tiled fp32 kernel already sits at ~94 % of what the chip delivers at its power-limited clock.
// bf16/f16 inputs alone cannot hold the 1e-3 max-abs budget through G3d's 21 stacked convs, so
// each fp32 operand is split into two f16 halves, v*S = hi + lo (|lo| <= 2^-11 |hi|, S a power of
// two that keeps lo out of the f16 subnormal range), and the product is accumulated in fp32 as
//     W*X ~= Whi*Xhi + Whi*Xlo + Wlo*Xhi            (dropped Wlo*Xlo term ~ 2^-22 relative)
// Every f16 x f16 product is exact in the MFMA's fp32 accumulate, so the result carries ~2^-21
// relative error per term — fp32 class — at 3 MFMAs of a 16x faster instruction (5.3x the fp32 rate).
//
// HBM layout is unchanged (fp32 NCDHW in, fp32 NCDHW out): activations are split while the halo
// tile is staged into LDS (registers -> cvt -> LDS, transposed to channel-contiguous 16-byte
// fragments); weights are split once at pack time into the exact LDS image of every
// (co tile, 16-channel chunk, 3-tap group) slab, so the weight stream is a plain lane-linear
// LDS-DMA (global_load_lds_dwordx4) of contiguous memory, double buffered.
//
// MFMA: v_mfma_f32_32x32x16_f16, A = weights [32 co x 16 ci], B = voxels [16 ci x 32 vox]
//   lane l holds 8 consecutive k (ci) of row/col (l&31):  k = 8*(l>>5) .. +7   (one 16-byte LDS read)
//   C/D: col j = lane&31 (voxel), row i = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (co)
#include "mphip_common.h"
#include "mphip_conv.h"

namespace mphip {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr float X_SCALE = 16.0f;          // activations: |x| < 4094 stays finite in f16; lo subnormal only below |x| ~ 8e-3
constexpr float F16_CLAMP = 65000.0f;
constexpr int F16X3_KC = 16;              // input channels per chunk = K of one MFMA
constexpr int F16X3_TG = 3;               // taps per weight slab
constexpr int F16X3_COT = 96;             // output channels per workgroup (3 MFMA row tiles)
constexpr int SLAB_HALFS = 2 * F16X3_TG * 2 * F16X3_COT * 8;  // [part][tap][kg][co][8] = 9216 halfs = 18432 B

__device__ __forceinline__ void split_f16(float v, _Float16 &hi, _Float16 &lo) {
    v = fminf(fmaxf(v, -F16_CLAMP), F16_CLAMP);
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

// ---- weight packing ----------------------------------------------------------------------------
// header (16 B): [0] inv_scale (float)  [1] scale (float)  [2] max|w| bits (uint)  [3] unused
__global__ void f16x3_absmax_kernel(const float *__restrict__ w, size_t n, unsigned *__restrict__ hdr) {
    float m = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(w[i]));
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(hdr + 2, __float_as_uint(m));  // non-negative floats order like uints
}

__device__ __forceinline__ float weight_scale(unsigned maxbits) {
    float m = __uint_as_float(maxbits);
    if (!(m > 0.0f) || !(m < 1e30f)) return 1.0f;
    int e;
    frexpf(m, &e);              // m = f * 2^e, f in [0.5,1)  ->  m < 2^e
    return ldexpf(1.0f, 15 - e);  // m*scale < 2^15 = 32768
}

// OIDHW [Co,Ci,3,3,3] fp32 -> slabs[(cot*nchunks + chunk)*9 + g][part][tap][kg][co][8] f16 (after the header)
__global__ void f16x3_pack_kernel(const float *__restrict__ w, _Float16 *__restrict__ out, const unsigned *hdr_in,
                                  float *__restrict__ hdr_out, int Co, int Ci) {
    const float scale = weight_scale(hdr_in[2]);
    const int nchunks = Ci / F16X3_KC;
    const size_t n = (size_t)(Co / F16X3_COT) * nchunks * 9 * (SLAB_HALFS / 2);  // one thread per (hi,lo) pair
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const int e = (int)(r % 8); r /= 8;
        const int co = (int)(r % F16X3_COT); r /= F16X3_COT;
        const int kg = (int)(r % 2); r /= 2;
        const int tg = (int)(r % F16X3_TG); r /= F16X3_TG;
        const int g = (int)(r % 9); r /= 9;
        const int chunk = (int)(r % nchunks);
        const int cot = (int)(r / nchunks);
        const int ci = chunk * F16X3_KC + kg * 8 + e, tap = g * F16X3_TG + tg, cog = cot * F16X3_COT + co;
        _Float16 hi, lo;
        split_f16(w[((size_t)cog * Ci + ci) * 27 + tap] * scale, hi, lo);
        const size_t slab = ((size_t)cot * nchunks + chunk) * 9 + g;
        const size_t inner = (((size_t)tg * 2 + kg) * F16X3_COT + co) * 8 + e;
        out[slab * SLAB_HALFS + inner] = hi;
        out[slab * SLAB_HALFS + SLAB_HALFS / 2 + inner] = lo;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        hdr_out[0] = 1.0f / scale;
        hdr_out[1] = scale;
    }
}

// ---- the conv kernel ---------------------------------------------------------------------------
template <int TD, int TH, int TW>
__global__ void __launch_bounds__(256)
conv3d_k3_f16x3_kernel(const float *__restrict__ x, const _Float16 *__restrict__ wslabs, const float *__restrict__ whdr,
                       const float *__restrict__ bias, float *__restrict__ y, int N, int Ci, int Co, int D, int H, int W,
                       int chunks_per_split, unsigned x_bytes) {
    constexpr int MT = 3, KC = F16X3_KC;
    constexpr int TVOX = TD * TH * TW;
    constexpr int NT = TVOX / 128;
    constexpr int HD = TD + 2, HH = TH + 2, HWp = TW + 2;
    constexpr int XV = HD * HH * HWp;                 // halo voxels
    constexpr int X_PART = 2 * XV * 8;                // halfs per part (hi or lo): [kg][vox][8]
    constexpr int XI = (8 * XV + 255) / 256;          // (channel pair, voxel) items per thread
    constexpr int W_BUF = SLAB_HALFS;                 // halfs per weight buffer
    constexpr int W_PIECES = SLAB_HALFS * 2 / 1024;   // 1-KiB DMA pieces per slab (18)
    __shared__ __attribute__((aligned(16))) _Float16 smem[2 * W_BUF + 2 * X_PART];
    _Float16 *const Ws = smem;               // [2 buffers][part][tap][kg][co][8]
    _Float16 *const Xs = smem + 2 * W_BUF;   // [part][kg][vox][8]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kg = lane >> 5;
    const int HW = H * W, DHW = D * HW;

    const int tiles_w = W / TW, tiles_h = H / TH, tiles_d = D / TD;
    int bid = blockIdx.x;
    const int tw = bid % tiles_w; bid /= tiles_w;
    const int th = bid % tiles_h; bid /= tiles_h;
    const int td = bid % tiles_d;
    const int n = bid / tiles_d;
    const int d0 = td * TD, h0 = th * TH, w0 = tw * TW;
    const int cot = blockIdx.y;
    const int nchunks = Ci / KC;
    const int c_begin = blockIdx.z * chunks_per_split;
    const int c_end = min(nchunks, c_begin + chunks_per_split);
    if (c_begin >= c_end) return;

    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)x_bytes, 0x00020000);

    // X staging plan: item e = i*256+tid -> (channel pair p = e / XV, halo voxel = e % XV)
    unsigned xsrc[XI];   // byte offset of (ci = 2p, voxel) for chunk 0, OOB when padding / beyond the item count
    int xdst[XI];        // half index in a part: ((kg*XV + vox)*8 + (p%4)*2), -1 when unused
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int e = i * 256 + tid;
        unsigned off = OOB;
        int dsti = -1;
        if (e < 8 * XV) {
            const int p = e / XV, r = e % XV;
            const int gd = d0 - 1 + r / (HH * HWp), gh = h0 - 1 + (r / HWp) % HH, gw = w0 - 1 + r % HWp;
            if ((unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W)
                off = (unsigned)((((long)n * Ci + 2 * p) * DHW + (long)gd * HW + gh * W + gw) * 4);
            dsti = (((p / 4) * XV + r) * 8) + (p % 4) * 2;
        }
        xsrc[i] = off;
        xdst[i] = dsti;
    }
    const unsigned chan_stride = (unsigned)DHW * 4u;

    float xr0[XI], xr1[XI];
#define F16X3_LOAD_X(chunk)                                                                       \
    {                                                                                             \
        const unsigned soff_ = (unsigned)((long)(chunk) * KC * DHW * 4);                          \
        _Pragma("unroll") for (int i = 0; i < XI; ++i) {                                          \
            xr0[i] = buf_load_f(rsrc, xsrc[i], soff_);                                            \
            xr1[i] = buf_load_f(rsrc, xsrc[i] == OOB ? OOB : xsrc[i] + chan_stride, soff_);       \
        }                                                                                         \
    }
#define F16X3_WRITE_X()                                                                           \
    {                                                                                             \
        _Pragma("unroll") for (int i = 0; i < XI; ++i) {                                          \
            if (xdst[i] >= 0) {                                                                   \
                _Float16 h0_, l0_, h1_, l1_;                                                      \
                split_f16(xr0[i] * X_SCALE, h0_, l0_);                                            \
                split_f16(xr1[i] * X_SCALE, h1_, l1_);                                            \
                half2v hv_ = {h0_, h1_}, lv_ = {l0_, l1_};                                        \
                *reinterpret_cast<half2v *>(Xs + xdst[i]) = hv_;                                  \
                *reinterpret_cast<half2v *>(Xs + X_PART + xdst[i]) = lv_;                         \
            }                                                                                     \
        }                                                                                         \
    }
    // weight slab (chunk, group) -> buffer: W_PIECES 1-KiB pieces, wave w takes pieces w, w+4, ...
#define F16X3_DMA_W(chunk, grp, wbuf)                                                             \
    {                                                                                             \
        const _Float16 *src_ = wslabs + (((size_t)cot * nchunks + (chunk)) * 9 + (grp)) * SLAB_HALFS + lane * 8; \
        _Pragma("unroll") for (int q = 0; q < (W_PIECES + 3) / 4; ++q) {                          \
            const int piece_ = q * 4 + wave;                                                      \
            if (piece_ < W_PIECES)                                                                \
                __builtin_amdgcn_global_load_lds(                                                 \
                    (const __attribute__((address_space(1))) void *)(src_ + piece_ * 512),        \
                    (__attribute__((address_space(3))) void *)(Ws + (wbuf) * W_BUF + piece_ * 512), 16, 0, 0); \
        }                                                                                         \
    }

    // fragment bases (halfs)
    const int a_base = (kg * F16X3_COT + j) * 8;           // + ((part*3 + tap)*2*96 + m*32)*8
    int b_base[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int v = (wave * NT + t) * 32 + j;
        const int vw = v % TW, vh = (v / TW) % TH, vd = v / (TW * TH);
        b_base[t] = (kg * XV + (vd * HH + vh) * HWp + vw) * 8;
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.0f;

    // prologue: X(chunk0) -> LDS, W(chunk0, group 0) -> buffer 0
    F16X3_DMA_W(c_begin, 0, 0);
    F16X3_LOAD_X(c_begin);
    F16X3_WRITE_X();
    __syncthreads();

    int wb = 0;
    for (int c = c_begin; c < c_end; ++c) {
        const bool more = c + 1 < c_end;
        if (more) F16X3_LOAD_X(c + 1);
#pragma unroll
        for (int g = 0; g < 9; ++g) {
            // stream the next slab while this one is consumed
            if (g < 8) {
                F16X3_DMA_W(c, g + 1, wb ^ 1);
            } else if (more) {
                F16X3_DMA_W(c + 1, 0, wb ^ 1);
            }
            const _Float16 *wsb = Ws + wb * W_BUF + a_base;
#pragma unroll
            for (int tg = 0; tg < F16X3_TG; ++tg) {
                const int tap = g * F16X3_TG + tg;
                const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
                const int toff = ((kd * HH + kh) * HWp + kw) * 8;
                half8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    ah[m] = *reinterpret_cast<const half8 *>(wsb + (tg * 2 * F16X3_COT + m * 32) * 8);
                    al[m] = *reinterpret_cast<const half8 *>(wsb + SLAB_HALFS / 2 + (tg * 2 * F16X3_COT + m * 32) * 8);
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    bh[t] = *reinterpret_cast<const half8 *>(Xs + b_base[t] + toff);
                    bl[t] = *reinterpret_cast<const half8 *>(Xs + X_PART + b_base[t] + toff);
                }
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh[t], acc[m][t], 0, 0, 0);
                        acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl[t], acc[m][t], 0, 0, 0);
                        acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh[t], acc[m][t], 0, 0, 0);
                    }
            }
            __syncthreads();  // slab (g+1) landed (DMA drained by the barrier's vmcnt(0)); slab g free
            wb ^= 1;
        }
        if (more) {
            F16X3_WRITE_X();  // every wave is past its last read of the X tile (barrier above)
            __syncthreads();
        }
    }
#undef F16X3_LOAD_X
#undef F16X3_WRITE_X
#undef F16X3_DMA_W

    const bool direct = gridDim.z == 1;
    float *dst = direct ? y : y + (size_t)blockIdx.z * N * Co * DHW;
    const float unscale = whdr[0] * (1.0f / X_SCALE);
    const int co0 = cot * F16X3_COT;
    float bv[MT][16];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
            bv[m][reg] = (direct && bias) ? bias[co0 + m * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kg] : 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int v = (wave * NT + t) * 32 + j;
        const int vw = v % TW, vh = (v / TW) % TH, vd = v / (TW * TH);
        float *dv = dst + (size_t)n * Co * DHW + (size_t)(d0 + vd) * HW + (h0 + vh) * W + w0 + vw;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int co = co0 + m * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kg;
                dv[(size_t)co * DHW] = acc[m][t][reg] * unscale + bv[m][reg];
            }
        }
    }
}

bool f16x3_supported(int N, int Ci, int Co, int D, int H, int W, int k) {
    return k == 3 && Ci % F16X3_KC == 0 && Co % F16X3_COT == 0 && H % 8 == 0 && W % 8 == 0 && D % 2 == 0 &&
           (size_t)N * Ci * D * H * W * 4 < 0x80000000ull;
}

size_t f16x3_packed_bytes(int Co, int Ci) {
    return 16 + (size_t)(Co / F16X3_COT) * (Ci / F16X3_KC) * 9 * SLAB_HALFS * sizeof(_Float16);
}

F16x3Plan f16x3_plan(int N, int Ci, int Co, int D, int H, int W) {
    F16x3Plan p;
    p.td = D % 4 == 0 ? 4 : 2;
    const long tiles = (long)N * (D / p.td) * (H / 8) * (W / 8);
    const int nchunks = Ci / F16X3_KC;
    int sp = 1;
    while (tiles * (Co / F16X3_COT) * sp < 512 && nchunks / (sp * 2) >= 3) sp *= 2;
    p.splits = sp;
    p.chunks_per_split = (nchunks + sp - 1) / sp;
    p.grid = dim3((unsigned)tiles, Co / F16X3_COT, sp);
    return p;
}

int f16x3_pack(const float *w, void *out, int Co, int Ci, hipStream_t s) {
    unsigned *hdr = (unsigned *)out;
    hipError_t e = hipMemsetAsync(out, 0, 16, s);
    if (e != hipSuccess) {
        set_error("pack_conv_weight(f16x3): memset: %s", hipGetErrorString(e));
        return MPHIP_ELAUNCH;
    }
    const size_t n = (size_t)Co * Ci * 27;
    hipLaunchKernelGGL(f16x3_absmax_kernel, dim3(512), dim3(256), 0, s, w, n, hdr);
    hipLaunchKernelGGL(f16x3_pack_kernel, dim3(2048), dim3(256), 0, s, w, (_Float16 *)((char *)out + 16), (const unsigned *)hdr,
                       (float *)out, Co, Ci);
    return check_launch("pack_conv_weight(f16x3)");
}

int f16x3_launch(const F16x3Plan &p, const float *x, const void *wpacked, const float *bias, float *dst, int N, int Ci,
                 int Co, int D, int H, int W, hipStream_t s) {
    const float *hdr = (const float *)wpacked;
    const _Float16 *slabs = (const _Float16 *)((const char *)wpacked + 16);
    const unsigned xb = (unsigned)((size_t)N * Ci * D * H * W * 4);
    if (p.td == 4)
        hipLaunchKernelGGL((conv3d_k3_f16x3_kernel<4, 8, 8>), p.grid, dim3(256), 0, s, x, slabs, hdr, bias, dst, N, Ci, Co, D, H,
                           W, p.chunks_per_split, xb);
    else
        hipLaunchKernelGGL((conv3d_k3_f16x3_kernel<2, 8, 8>), p.grid, dim3(256), 0, s, x, slabs, hdr, bias, dst, N, Ci, Co, D, H,
                           W, p.chunks_per_split, xb);
    return check_launch("conv3d_fwd(f16x3)");
}

}  // namespace mphip
