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
#include <stdlib.h>

#include <algorithm>

#include <hip/hip_ext.h>

#include "mphip_ablate.h"
#include "mphip_common.h"
#include "mphip_conv.h"
#include "mphip_f16x3.h"

namespace mphip {

constexpr float X_SCALE = 16.0f;          // legacy fixed activation scale: only when a caller passes no range descriptor
constexpr int F16X3_KC = 16;              // input channels per chunk = K of one MFMA
constexpr int F16X3_TG = 3;               // taps per packed weight slab
constexpr int F16X3_NG = 27 / F16X3_TG;   // slabs per 16-channel chunk
constexpr int F16X3_COT = 96;             // output channels per workgroup (3 MFMA row tiles)
constexpr int SLAB_HALFS = 2 * F16X3_TG * 2 * F16X3_COT * 8;  // [part][tap][kg][co][8] = 9216 halfs = 18432 B

// Range: every operand tensor carries a range descriptor (mphip_common.h) and is scaled by its own power of two before the
// split, so max|x|*scale < 2^14 and no finite value can leave the f16 range.  Whatever still does — a non-finite input, or a
// finite one beyond a WRONG descriptor handed in by a caller — is not clamped: hi takes the value itself (Inf/NaN in f16),
// lo = 0, and the MFMA propagates Inf/NaN into the output exactly like the reference's fp32 conv would.  Such elements are
// counted (one atomic per wavefront that saw any, i.e. none in normal operation): mphip_f16x3_saturation_count().
__device__ unsigned long long g_f16x3_saturated;

#ifdef MPHIP_PROFILE_PHASES
// dev instrumentation: cycles (s_memtime) per phase, summed over all waves: [0] prologue [1] X-load issue [2] DMA issue
// [3] tap loop (fragment reads + MFMAs) [4] wait at the group barrier [5] X write + its barrier [6] epilogue [7] waves
__device__ unsigned long long g_f16x3_prof[8];
#define PROF_DECL unsigned long long pt_ = __builtin_readcyclecounter(), pa_[7] = {0, 0, 0, 0, 0, 0, 0}
#define PROF_ADD(slot) { const unsigned long long n_ = __builtin_readcyclecounter(); pa_[slot] += n_ - pt_; pt_ = n_; }
#define PROF_FLUSH if ((threadIdx.x & 63) == 0) { for (int q_ = 0; q_ < 7; ++q_) atomicAdd(&g_f16x3_prof[q_], pa_[q_]); atomicAdd(&g_f16x3_prof[7], 1ull); }
#else
#define PROF_DECL
#define PROF_ADD(slot)
#define PROF_FLUSH
#endif

#define F16X3_SAT_COUNT(a_, b_) sat_ += !(fabsf((a_) * x_scale) <= F16_CLAMP) + !(fabsf((b_) * x_scale) <= F16_CLAMP);  /* NaN counts */

// ---- weight packing ----------------------------------------------------------------------------
// header (16 B): [0] inv_scale (float)  [1] scale (float)  [2] max|w| bits (uint)  [3] unused
__device__ __forceinline__ void f16x3_absmax_body(const float *__restrict__ w, size_t n, unsigned *__restrict__ hdr, unsigned bid, unsigned nblk) {
    float m = 0.0f;
    const size_t n4 = ((uintptr_t)w & 15) == 0 ? n / 4 : 0;  // 16-byte loads when the tensor is aligned (torch allocations are)
    const float4 *w4 = reinterpret_cast<const float4 *>(w);
    for (size_t i = (size_t)bid * blockDim.x + threadIdx.x; i < n4; i += (size_t)nblk * blockDim.x) {
        const float4 q = w4[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(q.x), fabsf(q.y))), fmaxf(fabsf(q.z), fabsf(q.w)));
    }
    for (size_t i = n4 * 4 + (size_t)bid * blockDim.x + threadIdx.x; i < n; i += (size_t)nblk * blockDim.x)
        m = fmaxf(m, fabsf(w[i]));
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s, 64));
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    // Same-address atomics serialise in L2 (with one per workgroup they, not the read, set this kernel's time — which is why the
    // grid used to be capped at 256 workgroups, i.e. 1/8 of the read bandwidth): a workgroup only issues its atomicMax when its
    // maximum beats what the header already holds (a device-scope load), so after the first few workgroups almost none do.
    if (threadIdx.x == 0) {
        const unsigned mine = __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));  // non-negative floats order like uints
        if (mine > __hip_atomic_load(hdr + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(hdr + 2, mine);
    }
}
__global__ void __launch_bounds__(256) f16x3_absmax_kernel(const float *__restrict__ w, size_t n, unsigned *__restrict__ hdr) {
    f16x3_absmax_body(w, n, hdr, blockIdx.x, gridDim.x);
}

// OIDHW [Co,Ci,3,3,3] fp32 -> slabs[(cot*nchunks + chunk)*NG + g][part][tap][kg][co][8] f16 (after the header).
// A training step re-packs every weight twice (forward and bwd-data direction: 2 x 194 MB read, 2 x 194 MB written for G3d), so
// this is a bandwidth kernel: a workgroup owns (32 output channels, one 16-channel chunk), loads that tile with 16-byte loads
// of contiguous runs (forward: 32 runs of 16 ci x 27 taps; transposed: 16 runs of 32 co x 27 taps), transposes it in LDS (odd
// pitches: conflict-free both ways) and writes every (slab, part, tap, kg) piece as 32 contiguous 16-byte fragments.
constexpr int PK_CO = 32;
constexpr int PK_LDS_FLOATS = PK_CO * (F16X3_KC * 27 + 1);   // forward: 32 rows of 433; transposed: 16 rows of 865 (fewer floats)
static_assert(PK_LDS_FLOATS >= F16X3_KC * (PK_CO * 27 + 1), "pack tile");
__device__ __forceinline__ void
f16x3_pack_body(const float *__restrict__ w, _Float16 *__restrict__ out, const unsigned *hdr_in, float *__restrict__ hdr_out, int Co,
                int Ci, int transposed, _Float16 *__restrict__ wino_out /* the F(2,3) kernel's slabs (conv3d_f16x3_wino.hip) or nullptr */,
                const int block /* of this weight's (Co/32) x (Ci/16) */, float *__restrict__ tile /* LDS, PK_LDS_FLOATS */) {
    const float scale = weight_scale(hdr_in[2]);
    const int nchunks = Ci / F16X3_KC, subs = F16X3_COT / PK_CO;
    int bid = block;
    const int sub = bid % subs; bid /= subs;
    const int chunk = bid % nchunks;
    const int cot = bid / nchunks;
    const int cog0 = cot * F16X3_COT + sub * PK_CO, ci0 = chunk * F16X3_KC;
    // transposed: w is the original conv's [Ci][Co][27] weight, this pack its bwd-data conv (taps reversed)
    const int run = transposed ? PK_CO * 27 : F16X3_KC * 27, pitch = run + 1;  // forward: 32 rows of 16 ci x 27; transposed: 16 rows of 32 co x 27
    constexpr int NQ = PK_CO * F16X3_KC * 27 / 4, NIT = (NQ + 255) / 256;  // 16-byte pieces of the tile; all loads issued up front
    float4 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = it * 256 + threadIdx.x;
        const int r = i / (run / 4), q = i % (run / 4);
        const size_t src = transposed ? ((size_t)(ci0 + r) * Co + cog0) * 27 : ((size_t)(cog0 + r) * Ci + ci0) * 27;
        if (i < NQ) v[it] = *reinterpret_cast<const float4 *>(w + src + 4 * q);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = it * 256 + threadIdx.x;
        if (i < NQ) {
            float *d = tile + (i / (run / 4)) * pitch + 4 * (i % (run / 4));
            d[0] = v[it].x; d[1] = v[it].y; d[2] = v[it].z; d[3] = v[it].w;
        }
    }
    __syncthreads();
    // one item = (slab g, tap tg, kg, co): 8 consecutive ci -> one 16-byte fragment of hi and one of lo
    for (int i = threadIdx.x; i < F16X3_NG * F16X3_TG * 2 * PK_CO; i += 256) {
        const int co = i % PK_CO;
        int r = i / PK_CO;
        const int kg = r % 2; r /= 2;
        const int tg = r % F16X3_TG;
        const int g = r / F16X3_TG;
        const int tap = g * F16X3_TG + tg;
        half8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ci = kg * 8 + e;
            const float v = transposed ? tile[ci * pitch + co * 27 + (26 - tap)] : tile[co * pitch + ci * 27 + tap];
            _Float16 h, l;
            split_f16(v * scale, h, l);
            hi[e] = h; lo[e] = l;
        }
        const size_t slab = ((size_t)cot * nchunks + chunk) * F16X3_NG + g;
        const size_t inner = (((size_t)tg * 2 + kg) * F16X3_COT + sub * PK_CO + co) * 8;
        *reinterpret_cast<half8 *>(out + slab * SLAB_HALFS + inner) = hi;
        *reinterpret_cast<half8 *>(out + slab * SLAB_HALFS + SLAB_HALFS / 2 + inner) = lo;
    }
    // The F(2,3) kernel's slabs from the SAME staged tile (training re-packs every weight every step: a second kernel that gathered the
    // 27-float-strided taps again took 0.49 ms of a 12 ms step): slabs[(cot*nchunks + chunk)*9 + (kd*3+kh)][part][position][kg][co][8],
    // filter transform u0 = g0, u1 = (g0+g1+g2)/2, u2 = (g0-g1+g2)/2, u3 = g2 along kw, exact in double, then the same split
    if (wino_out) {
        constexpr int WSLAB = 2 * 4 * 2 * F16X3_COT * 8;
        for (int i = threadIdx.x; i < 9 * 4 * 2 * PK_CO; i += 256) {
            const int co = i % PK_CO;
            int r = i / PK_CO;
            const int kg = r % 2; r /= 2;
            const int p = r % 4;
            const int g = r / 4;
            half8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ci = kg * 8 + e;
                double t[3];
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int tap = g * 3 + kw;
                    t[kw] = transposed ? tile[ci * pitch + co * 27 + (26 - tap)] : tile[co * pitch + ci * 27 + tap];
                }
                const double u = p == 0 ? t[0] : p == 1 ? 0.5 * (t[0] + t[1] + t[2]) : p == 2 ? 0.5 * (t[0] - t[1] + t[2]) : t[2];
                const double us = u * (double)scale;
                const _Float16 h = (_Float16)(float)us;
                hi[e] = h;
                lo[e] = (fabs(us) <= (double)F16_CLAMP) ? (_Float16)(float)(us - (double)(float)h) : (_Float16)0.0f;
            }
            const size_t slab = ((size_t)cot * nchunks + chunk) * 9 + g;
            const size_t inner = ((size_t)(p * 2 + kg) * F16X3_COT + sub * PK_CO + co) * 8;
            *reinterpret_cast<half8 *>(wino_out + slab * WSLAB + inner) = hi;
            *reinterpret_cast<half8 *>(wino_out + slab * WSLAB + WSLAB / 2 + inner) = lo;
        }
    }
    if (block == 0 && threadIdx.x == 0) {
        hdr_out[0] = 1.0f / scale;
        hdr_out[1] = scale;
    }
}
__global__ void __launch_bounds__(256)
f16x3_pack_kernel(const float *__restrict__ w, _Float16 *__restrict__ out, const unsigned *hdr_in, float *__restrict__ hdr_out, int Co,
                  int Ci, int transposed, _Float16 *__restrict__ wino_out) {
    __shared__ __attribute__((aligned(16))) float tile[PK_LDS_FLOATS];
    f16x3_pack_body(w, out, hdr_in, hdr_out, Co, Ci, transposed, wino_out, (int)blockIdx.x, tile);
}

// ---- the conv kernel ---------------------------------------------------------------------------
// GS = packed slabs (of 3 taps) streamed per barrier interval: 3 -> 4 barriers per chunk and 110 KB of
// weight buffers (one workgroup per CU), 1 -> 10 barriers per chunk and 37 KB.
// MTS = 32-channel row tiles of the 96-channel slab one workgroup computes: 3 (all; blockIdx.y = the 96-channel tile) or 1
// (blockIdx.y = 3 * tile + third: demand-driven launches, where a tile is ONE CU's matrix work and three CUs per tile cut its latency)
template <int TD, int TH, int TW, int NWAVES, int GS, int MTS = 3>
__device__ __forceinline__ void
conv3d_k3_f16x3_body(const float *__restrict__ x, const _Float16 *__restrict__ wslabs, const float *__restrict__ whdr,
                     const float *__restrict__ bias, float *__restrict__ y, int N, int Ci, int Co, int D, int H, int W,
                     int chunks_per_split, unsigned x_bytes, const float *__restrict__ in_affine, int in_relu,
                     const float *__restrict__ x_scale_p /* range descriptor of x */, int tiles_total, int xcd_aware,
                     const int *__restrict__ roi, int roi_frames, float *__restrict__ gn_part) {
    constexpr int MT = MTS, KC = F16X3_KC;
    static_assert(MTS == 3 || MTS == 1, "row tiles per workgroup");
    // operand scale of the input tensor: from its range descriptor (activations: max|x| noted by the producing kernel or
    // mphip_absmax_range; gradients: mphip_grad_prep) — per tensor, a power of two
    // Demand-driven evaluation (tile_list != nullptr): only the listed output tiles are computed — G3d's final_conv feeds
    // apply_warping_field + sum(dim=2) (model.py:1167-1171), a gather whose sample positions are known before the conv is
    // launched; on the reference's fields they cover a ~5^3 corner of the 16x64x64 volume, i.e. 2 of this
    // conv's 256 tiles per frame.  tile_list = {count, id, id, ...} (roi_tile_list_kernel below); the persistent workgroups
    // deal the LISTED tiles round-robin, so sixteen needed tiles run on sixteen CUs (walking the full id range and skipping
    // would leave them on the four workgroups whose stride-256 walks contain the low-corner tiles of every frame).
    const int *const tile_list = roi;
    const int ntiles = tile_list ? tile_list[0] : tiles_total;
    auto tile_at = [&](int j) -> int { return tile_list ? tile_list[1 + j] : j; };
    // XCD-aware placement over the WHOLE grid (r04): the hardware deals workgroup ids (x fastest, then y, z) round-robin over the 8 XCDs;
    // xcd_remap over the linear id gives every XCD a contiguous range of logical ids — the tiles of one (96-channel tile, K-split) pair first.  On the 2x8x8 level a pair's 8 tiles (one per frame) share 64 MB of packed weights that no L2 holds: with the
    // hardware's order they sat on 8 different XCDs and every XCD streamed all of it (343 MB from HBM per launch for 64 MB of weights,
    // 4.9 TB/s: the launches were HBM-bound, tools/pmc_step_traffic.sh); now each weight slab crosses the fabric once.
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    const unsigned ids = gridDim.x * gridDim.y * gridDim.z;
    if (!tile_list && xcd_aware) {
        const unsigned id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned lg = xcd_remap(id, ids);
        bx = lg % gridDim.x;
        by = (lg / gridDim.x) % gridDim.y;
        bz = lg / (gridDim.x * gridDim.y);
    }
    const int j_first = (int)bx;
    if (j_first >= ntiles) return;   // (workgroup-uniform, before any barrier)
    (void)roi_frames;
    float x_scale = X_SCALE, x_unscale = 1.0f / X_SCALE;
    if (x_scale_p) range_scale_block(x_scale_p, x_scale, x_unscale);  // (folds the producer's per-workgroup maxima; barriers inside)
    constexpr int TVOX = TD * TH * TW;
    constexpr int NTHR = NWAVES * 64;
    constexpr int NT = TVOX / (32 * NWAVES);          // 32-voxel column tiles per wave
    static_assert(NT >= 1 && TVOX % (32 * NWAVES) == 0, "tile/wave shape");
    constexpr int HD = TD + 2, HH = TH + 2, HWp = TW + 2;
    constexpr int XV = HD * HH * HWp;                 // halo voxels
    constexpr int X_PART = 2 * XV * 8;                // halfs per part (hi or lo): [kg][vox][8]
    constexpr int W_BUF = GS * SLAB_HALFS;            // halfs per weight buffer (GS consecutive slabs)
    constexpr int W_PIECES = W_BUF * 2 / 1024;        // 1-KiB DMA pieces per group (18 per slab)
    constexpr int NGRP = (F16X3_NG + GS - 1) / GS;    // barrier intervals per chunk (the last group may be shorter)
    constexpr int GT = GS * F16X3_TG;                 // taps per full interval
    constexpr int AFF_MAX_CI = 768;          // per-channel (scale, shift) of the fused input GroupNorm, kept in LDS
    __shared__ __attribute__((aligned(16))) _Float16 smem[2 * W_BUF + 2 * X_PART + AFF_MAX_CI * 4];
    _Float16 *const Ws = smem;               // [2 buffers][part][tap][kg][co][8]
    _Float16 *const Xs = smem + 2 * W_BUF;   // [part][kg][vox][8]
    float *const aff = reinterpret_cast<float *>(smem + 2 * W_BUF + 2 * X_PART);  // [Ci][2] (one array: see guide §5 trap (a))

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kg = lane >> 5;
    const int HW = H * W, DHW = D * HW;

    const int tiles_w = W / TW, tiles_h = H / TH, tiles_d = D / TD;
    // Persistent over tiles: gridDim.x is sized to what the chip holds at once and a workgroup walks tiles blockIdx.x,
    // +gridDim.x, ... — the output stores of one tile drain while the next tile's weight DMA and halo loads are already
    // in flight, and the per-workgroup launch / teardown gaps disappear (phase timing: prologue + epilogue were 15-25 %
    // of a wave's time with one tile per workgroup).
    int n, d0, h0, w0;
    auto decode_tile = [&](int tile_id) {
        int bid = tile_id;
        const int tw = bid % tiles_w; bid /= tiles_w;
        const int th = bid % tiles_h; bid /= tiles_h;
        const int td = bid % tiles_d;
        n = bid / tiles_d;
        d0 = td * TD; h0 = th * TH; w0 = tw * TW;
    };
    // XCD-aware start: workgroup ids go round-robin over the 8 XCDs, so consecutive ids get consecutive RANGES of tiles —
    // neighbouring tiles (which share halo rows) then run on the same XCD and meet in its L2
    decode_tile(tile_at(j_first));
    const int cot = MTS == 3 ? by : by / 3, mb = MTS == 3 ? 0 : by % 3;
    const int nchunks = Ci / KC;
    const int c_begin = bz * chunks_per_split;
    const int c_end = min(nchunks, c_begin + chunks_per_split);
    if (c_begin >= c_end) return;

    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)x_bytes, 0x00020000);

    // X staging, by halo ROW: a row is [left edge][TW interior voxels][right edge] of one channel.  An interior item is
    // (channel pair p, row, quad q): two 16-byte loads (channels 2p, 2p+1; voxels 4q..4q+3 — aligned, always inside the
    // volume in w) instead of eight 4-byte ones; edge items are the two single voxels.  A row outside the volume in d or h
    // is all padding: its loads get the out-of-range offset and return 0.  (12 load instructions per thread and chunk
    // instead of 68; +3 % on the full-resolution layers.)  Offsets are recomputed when needed instead of living in registers.
    const unsigned chan_stride = (unsigned)DHW * 4u;
    long nbase = 0;
    constexpr int ROWS = HD * HH;
    constexpr int QPR = TW / 4;                        // interior quads per row
    constexpr int NQ = 8 * ROWS * QPR;                 // interior items per chunk
    constexpr int QI = (NQ + NTHR - 1) / NTHR;
    constexpr int NE = 8 * ROWS * 2;                   // edge items per chunk
    constexpr int EI = (NE + NTHR - 1) / NTHR;
    auto row_src = [&](int row) -> unsigned {          // byte offset of (n, channel 0, row, w0) or OOB
        const int gd = d0 - 1 + row / HH, gh = h0 - 1 + row % HH;
        if ((unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H) return (unsigned)((nbase + (long)gd * HW + gh * W + w0) * 4);
        return OOB;
    };
    const bool fuse_in = in_affine != nullptr;  // block-uniform
    int aff_n = -1;

    f32x4 xq0[QI], xq1[QI];
    float xe0[EI], xe1[EI];
#define F16X3_LOAD_X(chunk)                                                                       \
    {                                                                                             \
        const unsigned soff_ = (unsigned)((long)(chunk) * KC * DHW * 4);                          \
        int tid_ = tid;                                                                           \
        asm volatile("" : "+v"(tid_)); /* opaque: keeps the plan out of registers across the K loop */ \
        _Pragma("unroll") for (int i = 0; i < QI; ++i) {                                          \
            const int e_ = i * NTHR + tid_;                                                       \
            const int p_ = e_ / (ROWS * QPR), rem_ = e_ % (ROWS * QPR);                           \
            unsigned o_ = e_ < NQ ? row_src(rem_ / QPR) : OOB;                                    \
            if (o_ != OOB) o_ += (unsigned)(2 * p_) * chan_stride + (unsigned)(rem_ % QPR) * 16u; \
            xq0[i] = buf_load_f4(rsrc, o_, soff_);                                                \
            xq1[i] = buf_load_f4(rsrc, o_ == OOB ? OOB : o_ + chan_stride, soff_);                \
        }                                                                                         \
        _Pragma("unroll") for (int i = 0; i < EI; ++i) {                                          \
            const int e_ = i * NTHR + tid_;                                                       \
            const int p_ = e_ / (ROWS * 2), rem_ = e_ % (ROWS * 2), side_ = rem_ & 1;             \
            unsigned o_ = e_ < NE ? row_src(rem_ >> 1) : OOB;                                     \
            const bool in_w_ = side_ ? w0 + TW < W : w0 > 0;                                      \
            o_ = (o_ != OOB && in_w_) ? o_ + (unsigned)(2 * p_) * chan_stride + (side_ ? TW * 4 : -4) : OOB; \
            xe0[i] = buf_load_f(rsrc, o_, soff_);                                                 \
            xe1[i] = buf_load_f(rsrc, o_ == OOB ? OOB : o_ + chan_stride, soff_);                 \
        }                                                                                         \
    }
    /* one (channel pair, voxel) -> LDS: the preceding GroupNorm (+ReLU) if fused, scale, split, two b32 writes */ \
#define F16X3_PUT(chunk, p_, r_, va_, vb_, valid_)                                                \
    {                                                                                             \
        float v0_ = (va_), v1_ = (vb_);                                                           \
        if (fuse_in && (valid_)) {                                                                \
            const float4 sc_ = *reinterpret_cast<const float4 *>(aff + ((chunk) * KC + 2 * (p_)) * 2); \
            v0_ = v0_ * sc_.x + sc_.y;                                                            \
            v1_ = v1_ * sc_.z + sc_.w;                                                            \
            if (in_relu) {                                                                        \
                v0_ = fmaxf(v0_, 0.0f);                                                           \
                v1_ = fmaxf(v1_, 0.0f);                                                           \
            }                                                                                     \
        }                                                                                         \
        const int dst_ = ((((p_) / 4) * XV + (r_)) * 8) + ((p_) % 4) * 2;                         \
        _Float16 h0_, l0_, h1_, l1_;                                                              \
        F16X3_SAT_COUNT(v0_, v1_)                                                                 \
        split_f16(v0_ * x_scale, h0_, l0_);                                                       \
        split_f16(v1_ * x_scale, h1_, l1_);                                                       \
        half2v hv_ = {h0_, h1_}, lv_ = {l0_, l1_};                                                \
        *reinterpret_cast<half2v *>(Xs + dst_) = hv_;                                             \
        *reinterpret_cast<half2v *>(Xs + X_PART + dst_) = lv_;                                    \
    }
#define F16X3_WRITE_X(chunk)                                                                      \
    {                                                                                             \
        _Pragma("unroll") for (int i = 0; i < QI; ++i) {                                          \
            const int e_ = i * NTHR + tid + tz;                                                   \
            if (e_ < NQ) {                                                                        \
                const int p_ = e_ / (ROWS * QPR), rem_ = e_ % (ROWS * QPR), row_ = rem_ / QPR;    \
                const int r_ = row_ * HWp + 1 + (rem_ % QPR) * 4;                                 \
                const bool ok_ = fuse_in && row_src(row_) != OOB;                                 \
                _Pragma("unroll") for (int k = 0; k < 4; ++k) F16X3_PUT(chunk, p_, r_ + k, xq0[i][k], xq1[i][k], ok_) \
            }                                                                                     \
        }                                                                                         \
        _Pragma("unroll") for (int i = 0; i < EI; ++i) {                                          \
            const int e_ = i * NTHR + tid + tz;                                                   \
            if (e_ < NE) {                                                                        \
                const int p_ = e_ / (ROWS * 2), rem_ = e_ % (ROWS * 2), side_ = rem_ & 1, row_ = rem_ >> 1; \
                const int r_ = row_ * HWp + (side_ ? HWp - 1 : 0);                                \
                const bool ok_ = fuse_in && row_src(row_) != OOB && (side_ ? w0 + TW < W : w0 > 0); \
                F16X3_PUT(chunk, p_, r_, xe0[i], xe1[i], ok_)                                     \
            }                                                                                     \
        }                                                                                         \
    }
    // weight slab (chunk, group) -> buffer: W_PIECES 1-KiB pieces, wave w takes pieces w, w+4, ...
#define F16X3_DMA_W(chunk, grp, wbuf)                                                             \
    {                                                                                             \
        const _Float16 *src_ = wslabs + (((size_t)cot * nchunks + (chunk)) * F16X3_NG + (grp) * GS) * SLAB_HALFS + lane * 8; \
        const int npieces_ = (((grp) + 1) * GS <= F16X3_NG ? GS : F16X3_NG - (grp) * GS) * (SLAB_HALFS * 2 / 1024); \
        _Pragma("unroll") for (int q = 0; q < (W_PIECES + NWAVES - 1) / NWAVES; ++q) {            \
            const int piece_ = q * NWAVES + wave;                                                 \
            if (piece_ < npieces_) F16X3_DMA16(src_ + piece_ * 512, Ws + (wbuf) * W_BUF + piece_ * 512) \
        }                                                                                         \
    }
    // (the DMA is issued by hand, mphip_f16x3.h: with the builtin hipcc puts `s_waitcnt vmcnt(0)` in front of the first fragment read
    //  that follows a transfer — it cannot prove the read does not alias the destination — so every barrier interval opened with a
    //  full L2 round trip; now a transfer has the whole interval to land and is waited for right before the interval's barrier)
#ifdef MPHIP_BUILTIN_DMA   /* dev: same-box A/B against the compiler-issued DMA */
#define F16X3_DMA16(src_p_, dst_p_)                                                               \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src_p_),    \
                                     (__attribute__((address_space(3))) void *)(dst_p_), 16, 0, 0);
#define F16X3_DMA_LANDED()
#else
#define F16X3_DMA16(src_p_, dst_p_) lds_dma16((src_p_), (unsigned)(uintptr_t)(__attribute__((address_space(3))) _Float16 *)(dst_p_));
#define F16X3_DMA_LANDED() lds_dma_wait<0>();
#endif

    // ds_read_b128 is serviced in 16-lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} (per half-wave):
    // give every group two whole 8-voxel rows (2 x 128 contiguous bytes) instead of row fragments of
    // four different rows, which removes the 2-3 way bank conflicts of the natural j -> voxel order.
    const int jg = ((j >> 2) & 1) ^ ((j >> 3) & 1) ^ ((j >> 4) & 1);  // hardware group of lane j: 0 or 1
    const int jpos = j < 4 ? j : j < 12 ? j - 4 : j < 20 ? j - 8 : j < 28 ? j - 12 : j - 16;
    const int jv = jg * 16 + jpos;  // column (voxel slot) of lane j inside its 32-voxel tile
    // fragment bases (halfs)
    const int a_base = (kg * F16X3_COT + j + mb * 32) * 8;   // + ((part*TG + tap)*2*96 + m*32)*8
    // column slot -> voxel of the tile.  8-wide tiles with one column tile per wave (the 2x8x8 and 4x8x8 instantiations): a wave owns four rows
    // of a plane, and the two rows of a 16-lane read group are FOUR rows apart — 4 x 160 B = 32 banks (mod 64) between their 128-byte
    // windows, which therefore fill the 64 banks exactly.  With neighbouring rows (r03-r04) the windows overlapped on 8 banks: 34 % of the
    // 2x8x8 level's LDS cycles were bank conflicts (tools/pmc_step_lds.sh).  Any bijection works: slots only name accumulator columns.
    auto slot_voxel = [&](int t, int &vd, int &vh, int &vw) {
        if (TW == 8 && NT == 1 && TH == 8) {
            const int q = jv >> 3;
            vw = jv & 7;
            vd = wave / 2;
            vh = (wave % 2) * 2 + (q & 1) * 4 + (q >> 1);   // rows {A, A+4} (slots 0-15), {A+1, A+5} (slots 16-31), A = 0 or 2
        } else {
            const int v = (wave * NT + t) * 32 + jv;
            vw = v % TW; vh = (v / TW) % TH; vd = v / (TW * TH);
        }
    };
    int b_base[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int vd, vh, vw;
        slot_voxel(t, vd, vh, vw);
        b_base[t] = (kg * XV + (vd * HH + vh) * HWp + vw) * 8;
    }

    PROF_DECL;
    // first tile's prologue: X(chunk0) -> LDS, W(chunk0, group 0) -> buffer 0.  Later tiles find theirs already staged:
    // the last chunk of a tile loads the NEXT tile's first halo chunk and weight slab (cross-tile software pipeline).
    auto load_aff = [&]() {
        // the GroupNorm (+ReLU) that precedes this conv (model.py:506-507 -> 517 -> 518) is applied while the halo
        // tile is staged: x' = relu(x*scale[n,c] + shift[n,c]) inside the volume, 0 in the padding
        for (int i = tid; i < Ci * 2; i += NTHR) aff[i] = in_affine[(size_t)n * Ci * 2 + i];
        aff_n = n;
    };
    nbase = (long)n * Ci * DHW;
    if (fuse_in) {
        load_aff();
        __syncthreads();
    }
    int tz = 0;
    unsigned sat_ = 0;  // halo elements that left the f16 range (counted once per staging; halos overlap between tiles)
    F16X3_DMA_W(c_begin, 0, 0);
    F16X3_LOAD_X(c_begin);
    F16X3_WRITE_X(c_begin);
    F16X3_DMA_LANDED()
    __syncthreads();
    PROF_ADD(0)
    int wb = 0;
  for (int tj = j_first; tj < ntiles; tj += (int)gridDim.x) {
    const int en = n, ed0 = d0, eh0 = h0, ew0 = w0;  // this tile's coordinates (the staging variables move on to the next tile)
    const int etile = tile_at(tj);
    const bool has_next = tj + (int)gridDim.x < ntiles;
    asm volatile("" : "+v"(tz));  // opaque 0, new per tile: keeps per-tile-invariant index math / bias loads from being hoisted into registers
    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.0f;

    for (int c = c_begin; c < c_end; ++c) {
        const bool more = c + 1 < c_end;
        if (more) {
            F16X3_LOAD_X(c + 1);
        } else if (has_next) {
            decode_tile(tile_at(tj + (int)gridDim.x));  // staging now addresses the next tile (the epilogue uses en, ed0, ...)
            nbase = (long)n * Ci * DHW;
            // (the previous affine table was last read by the WRITE_X that ended the previous chunk, a barrier ago)
            if (fuse_in && n != aff_n) load_aff();
            F16X3_LOAD_X(c_begin);
        }
        PROF_ADD(1)
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
            // stream the next group of slabs while this one is consumed
            if (g < NGRP - 1) {
                F16X3_DMA_W(c, g + 1, wb ^ 1);
            } else if (more) {
                F16X3_DMA_W(c + 1, 0, wb ^ 1);
            } else if (has_next) {
                F16X3_DMA_W(c_begin, 0, wb ^ 1);  // the next tile's first slab (weights do not depend on the tile)
            }
            PROF_ADD(2)
            const _Float16 *wsb = Ws + wb * W_BUF + a_base;
            const int gt = ((g + 1) * GS <= F16X3_NG ? GS : F16X3_NG - g * GS) * F16X3_TG;  // taps in this group
#define F16X3_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x16_f16(a_, b_, c_, 0, 0, 0)
#ifndef MPHIP_F16X3_OLD_FRAGS
            // Fragment schedule: per tap the three products run as  P1 = Wlo*Xhi,  P2 = Whi*Xhi,  P3 = Whi*Xlo  (6 MFMAs = 192
            // MFMA cycles each).  Only the Xhi fragments are double buffered; every other fragment is loaded into the registers
            // its predecessor vacated one or two phases earlier:
            //   before P1(t): Whi(t), Xlo(t)  [free since P3(t-1)]  and Xhi(t+1) [other buffer]     needed at P2 / P3 / P1(t+1)
            //   before P2(t): Wlo(t+1)        [free since P1(t)]                                     needed at P1(t+1)
            // so every ds_read_b128 has >= 192 MFMA cycles of cover and the fragments take 48 registers instead of the 80 of two
            // full sets (which pushed this kernel into scratch spills: 256 VGPRs + 34 spilled, r01).
            half8 ah[MT], al[MT], bl[NT], bh[2][NT];
#define F16X3_TOFF(tg_) ((((g * GT + (tg_)) / 9) * HH + ((g * GT + (tg_)) / 3) % 3) * HWp + (g * GT + (tg_)) % 3) * 8
#define F16X3_WOFF(tg_) (((tg_) / F16X3_TG) * SLAB_HALFS + (((tg_) % F16X3_TG) * 2 * F16X3_COT) * 8)
#define F16X3_LD_AH(tg_) _Pragma("unroll") for (int m = 0; m < MT; ++m) ah[m] = *reinterpret_cast<const half8 *>(wsb + F16X3_WOFF(tg_) + m * 32 * 8);
#define F16X3_LD_AL(tg_) _Pragma("unroll") for (int m = 0; m < MT; ++m) al[m] = *reinterpret_cast<const half8 *>(wsb + F16X3_WOFF(tg_) + SLAB_HALFS / 2 + m * 32 * 8);
#define F16X3_LD_BH(buf_, tg_) _Pragma("unroll") for (int t = 0; t < NT; ++t) bh[buf_][t] = *reinterpret_cast<const half8 *>(Xs + b_base[t] + F16X3_TOFF(tg_));
#define F16X3_LD_BL(tg_) _Pragma("unroll") for (int t = 0; t < NT; ++t) bl[t] = *reinterpret_cast<const half8 *>(Xs + X_PART + b_base[t] + F16X3_TOFF(tg_));
            F16X3_LD_AL(0)
            F16X3_LD_BH(0, 0)
#pragma unroll
            for (int tg = 0; tg < GT; ++tg) {
                if (tg < gt) {
                    const int cur = tg & 1;
                    F16X3_LD_AH(tg)
                    F16X3_LD_BL(tg)
                    if (tg + 1 < gt) { F16X3_LD_BH(cur ^ 1, tg + 1) }
                    __builtin_amdgcn_sched_barrier(0);  // the loads above stay above this tap's MFMAs
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[m][t] = F16X3_MFMA(al[m], bh[cur][t], acc[m][t]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (tg + 1 < gt) { F16X3_LD_AL(tg + 1) }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[m][t] = F16X3_MFMA(ah[m], bh[cur][t], acc[m][t]);
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[m][t] = F16X3_MFMA(ah[m], bl[t], acc[m][t]);
                }
            }
#undef F16X3_TOFF
#undef F16X3_WOFF
#undef F16X3_LD_AH
#undef F16X3_LD_AL
#undef F16X3_LD_BH
#undef F16X3_LD_BL
#else
            // (r01 schedule, kept for same-box A/B: two full fragment sets, all of tap tg+1 loaded before the MFMAs of tap tg)
            half8 ah[2][MT], al[2][MT], bh[2][NT], bl[2][NT];
#define F16X3_LOAD_FRAGS(set, tg_)                                                                        \
    {                                                                                                     \
        const int tap_ = g * GT + (tg_);                                                                  \
        const int toff_ = (((tap_ / 9) * HH + (tap_ / 3) % 3) * HWp + tap_ % 3) * 8;                      \
        const int woff_ = ((tg_) / F16X3_TG) * SLAB_HALFS + (((tg_) % F16X3_TG) * 2 * F16X3_COT) * 8;     \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                  \
            ah[set][m] = *reinterpret_cast<const half8 *>(wsb + woff_ + m * 32 * 8);                      \
            al[set][m] = *reinterpret_cast<const half8 *>(wsb + woff_ + SLAB_HALFS / 2 + m * 32 * 8);     \
        }                                                                                                 \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                                  \
            bh[set][t] = *reinterpret_cast<const half8 *>(Xs + b_base[t] + toff_);                        \
            bl[set][t] = *reinterpret_cast<const half8 *>(Xs + X_PART + b_base[t] + toff_);               \
        }                                                                                                 \
    }
            F16X3_LOAD_FRAGS(0, 0);
#pragma unroll
            for (int tg = 0; tg < GT; ++tg) {
                if (tg < gt) {
                const int cur = tg & 1;
                if (tg + 1 < gt) F16X3_LOAD_FRAGS(cur ^ 1, tg + 1);
                __builtin_amdgcn_sched_barrier(0);  // keep the prefetch above this tap's MFMAs
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[m][t] = F16X3_MFMA(al[cur][m], bh[cur][t], acc[m][t]);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[m][t] = F16X3_MFMA(ah[cur][m], bl[cur][t], acc[m][t]);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[m][t] = F16X3_MFMA(ah[cur][m], bh[cur][t], acc[m][t]);
                }
            }
#undef F16X3_LOAD_FRAGS
#endif
#undef F16X3_MFMA
            PROF_ADD(3)
            F16X3_DMA_LANDED()
            __syncthreads();  // slab (g+1) landed (this wave's pieces: the wait above; the others': the barrier); slab g free
            PROF_ADD(4)
            wb ^= 1;
        }
        if (more) {
            F16X3_WRITE_X(c + 1);  // every wave is past its last read of the X tile (barrier above)
            __syncthreads();
        } else if (has_next) {
            F16X3_WRITE_X(c_begin);  // the next tile's first halo chunk; its barrier doubles as the next tile's "prologue done"
            __syncthreads();
        }
        PROF_ADD(5)
    }
#undef F16X3_WRITE_X
#undef F16X3_PUT
#undef F16X3_DMA_W

    const bool direct = gridDim.z == 1;
    float *dst = direct ? y : y + (size_t)bz * N * Co * DHW;
    const float unscale = whdr[0] * x_unscale;
    const int co0 = cot * F16X3_COT + mb * 32;
    if (gn_part) {
        // GroupNorm statistics of THIS conv's output without a pass over it: per-channel (sum, sum of squares) of the values about to
        // be stored, as raw accumulators (the finalize kernel applies unscale and the bias in double), over the wave's 32*NT voxels -> gn_part[co][tile][wave][2]; gn_tile_finalize_kernel (norm.hip) folds a frame's
        // tiles in double.  A register holds one channel at 32 voxel lanes (x 2 channel halves): registers 2p / 2p+1 trade 16-lane
        // rows (v_permlane16_swap: one add halves the pair), then four rotate-adds inside a row — 6 VALU ops per channel and
        // quantity (r02's version shuffled every register through the LDS crossbar, 960 ds_bpermute per tile and wave: -1.2 %).
        const size_t gn_rows = (size_t)tiles_total * NWAVES;   // channel-major [Co][tile * NWAVES + wave][2]: the finalize kernel reads rows of it
        float *gp = gn_part + ((size_t)co0 * gn_rows + (size_t)etile * NWAVES + wave) * 2;
        if (etile == 0 && tid == 0) gn_part[(size_t)tiles_total * NWAVES * Co * 2] = unscale;   // (behind the partials; same value from every co tile)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float s2[8], q2[8];
#pragma unroll
            for (int pr = 0; pr < 8; ++pr) {
                float sv[2], qv[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int reg = 2 * pr + h;
                    sv[h] = acc[m][0][reg];   // raw accumulators: the finalize kernel applies the operand unscale (a power of two) too
                    qv[h] = acc[m][0][reg] * acc[m][0][reg];
#pragma unroll
                    for (int t = 1; t < NT; ++t) {
                        sv[h] += acc[m][t][reg];
                        qv[h] = __builtin_fmaf(acc[m][t][reg], acc[m][t][reg], qv[h]);
                    }
                }
                typedef unsigned u2_ __attribute__((ext_vector_type(2)));
                const u2_ rs = __builtin_amdgcn_permlane16_swap(__float_as_uint(sv[0]), __float_as_uint(sv[1]), false, false);
                const u2_ rq = __builtin_amdgcn_permlane16_swap(__float_as_uint(qv[0]), __float_as_uint(qv[1]), false, false);
                s2[pr] = __uint_as_float(rs[0]) + __uint_as_float(rs[1]);   // rows 0/2: register 2p over voxel pairs, rows 1/3: register 2p+1
                q2[pr] = __uint_as_float(rq[0]) + __uint_as_float(rq[1]);
#define F16X3_ROW_ADD(v_, ctrl_) v_ += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v_), ctrl_, 0xf, 0xf, false));
                F16X3_ROW_ADD(s2[pr], 0x128) F16X3_ROW_ADD(q2[pr], 0x128)   // row_ror:8, :4, :2, :1 -> lane 0 of a row: the row's sum
                F16X3_ROW_ADD(s2[pr], 0x124) F16X3_ROW_ADD(q2[pr], 0x124)
                F16X3_ROW_ADD(s2[pr], 0x122) F16X3_ROW_ADD(q2[pr], 0x122)
                F16X3_ROW_ADD(s2[pr], 0x121) F16X3_ROW_ADD(q2[pr], 0x121)
#undef F16X3_ROW_ADD
            }
            if ((lane & 15) == 0) {
#pragma unroll
                for (int pr = 0; pr < 8; ++pr) {
                    const int reg = 2 * pr + ((lane >> 4) & 1);
                    const int col = m * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kg;
                    *reinterpret_cast<float2 *>(gp + (size_t)col * gn_rows * 2) = make_float2(s2[pr], q2[pr]);
                }
            }
        }
    }
    // (one 32-channel row tile at a time: its 16 bias values are loaded right before its stores — all 48 up front cost the 512-voxel
    //  instantiation 16 of its spilled registers)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        float bv[16];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
            bv[reg] = (direct && bias) ? bias[co0 + m * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kg + tz] : 0.0f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            int vd, vh, vw;
            slot_voxel(t, vd, vh, vw);
            float *dv = dst + (size_t)en * Co * DHW + (size_t)(ed0 + vd) * HW + (eh0 + vh) * W + ew0 + vw;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int co = co0 + m * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kg + tz;
                dv[(size_t)co * DHW] = acc[m][t][reg] * unscale + bv[reg];
            }
        }
    }
    PROF_ADD(6)
  }  // tiles
#undef F16X3_LOAD_X
#undef F16X3_WRITE_X
#undef F16X3_PUT
#undef F16X3_DMA_W
#undef F16X3_DMA16
#undef F16X3_DMA_LANDED
    if (__builtin_amdgcn_ballot_w64(sat_ != 0) != 0) {  // never taken in normal operation
        unsigned tot = sat_;
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) tot += __shfl_xor(tot, sft, 64);
        if (lane == 0) atomicAdd(&g_f16x3_saturated, (unsigned long long)tot);
    }
    PROF_FLUSH
}

// Demand-driven conv: the ids of the output tiles (TD x TH x TW voxels, id = ((n*tiles_d + td)*tiles_h + th)*tiles_w + tw) that
// one of the sample boxes {lx,ly,lz,ex,ey,ez,-,-} touches -> list = {count, id, ...}.  roi_frames == 0: box n belongs to frame n;
// > 0: the single frame serves that many boxes.  One workgroup; the order of the ids is irrelevant (tiles are independent).
__global__ void __launch_bounds__(1024)
roi_tile_list_kernel(const int *__restrict__ roi, int roi_frames, int tiles_total, int D, int H, int W, int TD, int TH, int TW,
                     int dilate, int *__restrict__ list) {
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const int tiles_w = W / TW, tiles_h = H / TH, tiles_d = D / TD;
    for (int t = threadIdx.x; t < tiles_total; t += 1024) {
        int bid = t;
        const int w0 = (bid % tiles_w) * TW; bid /= tiles_w;
        const int h0 = (bid % tiles_h) * TH; bid /= tiles_h;
        const int d0 = (bid % tiles_d) * TD;
        const int n = bid / tiles_d;
        const int first = roi_frames > 0 ? 0 : n, count = roi_frames > 0 ? roi_frames : 1;
        bool need = false;
        for (int f = first; f < first + count && !need; ++f) {
            const int *b = roi + f * 8;   // (dilate: the box grown by that many voxels on every side — bwd-data of a box of gradients)
            need = w0 < b[0] + b[3] + dilate && w0 + TW > b[0] - dilate && h0 < b[1] + b[4] + dilate && h0 + TH > b[1] - dilate &&
                   d0 < b[2] + b[5] + dilate && d0 + TD > b[2] - dilate;
        }
        if (need) list[1 + atomicAdd(&cnt, 1)] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) list[0] = cnt;
}

template <int TD, int TH, int TW, int NWAVES, int GS>
__global__ void __launch_bounds__(NWAVES * 64) __attribute__((amdgpu_waves_per_eu(2, 2)))  // 256 registers: two waves per SIMD
conv3d_k3_f16x3_kernel(const float *__restrict__ x, const _Float16 *__restrict__ wslabs, const float *__restrict__ whdr,
                       const float *__restrict__ bias, float *__restrict__ y, int N, int Ci, int Co, int D, int H, int W,
                       int chunks_per_split, unsigned x_bytes, const float *__restrict__ in_affine, int in_relu,
                       const float *__restrict__ x_scale_p, int tiles_total, int xcd_aware, const int *__restrict__ roi, int roi_frames,
                       float *__restrict__ gn_part) {
    conv3d_k3_f16x3_body<TD, TH, TW, NWAVES, GS>(x, wslabs, whdr, bias, y, N, Ci, Co, D, H, W, chunks_per_split, x_bytes, in_affine,
                                                 in_relu, x_scale_p, tiles_total, xcd_aware, roi, roi_frames, gn_part);
}

// demand-driven launches: one 32-channel third of a tile per workgroup (three CUs per tile; same bits: a row tile's accumulation
// does not depend on its neighbours)
template <int TD, int TH, int TW, int NWAVES, int GS>
__global__ void __launch_bounds__(NWAVES * 64) __attribute__((amdgpu_waves_per_eu(2, 2)))
conv3d_k3_f16x3_third_kernel(const float *__restrict__ x, const _Float16 *__restrict__ wslabs, const float *__restrict__ whdr,
                             const float *__restrict__ bias, float *__restrict__ y, int N, int Ci, int Co, int D, int H, int W,
                             int chunks_per_split, unsigned x_bytes, const float *__restrict__ in_affine, int in_relu,
                             const float *__restrict__ x_scale_p, int tiles_total, int xcd_aware, const int *__restrict__ roi, int roi_frames,
                             float *__restrict__ gn_part) {
    conv3d_k3_f16x3_body<TD, TH, TW, NWAVES, GS, 1>(x, wslabs, whdr, bias, y, N, Ci, Co, D, H, W, chunks_per_split, x_bytes, in_affine,
                                                    in_relu, x_scale_p, tiles_total, xcd_aware, roi, roi_frames, gn_part);
}

// (r02-r03 experiments removed in r04, measured and rejected — DESIGN.md 3: four "wide" waves, one per SIMD with 512 registers and 96 x 128
//  accumulators, 4-10 % slower; two 4-wave workgroups per CU on (4,8,8) tiles, +-0.  Both instantiations spilled 80-110 registers.)

// ---- k = 1: the 1x1x1 shortcut convs of G3d (model.py:510) on the same split-f16 arithmetic ---------------------------------
// Y[co][vox] = sum_ci W[co][ci] * X[ci][vox] is a plain GEMM that streams X once: HBM-bound.  No LDS, no barriers: a wave owns
// 64 voxels x 96 output channels (3 x 2 MFMA tiles, 96 accumulator registers); per 16-channel chunk a lane fetches its 8
// channels of 2 voxels straight from NCDHW (8 coalesced 128-byte row loads per tile), splits them in registers, and reads the
// 6 weight fragments (hi/lo x 3 row tiles, 16 B per lane) from the packed slab in L2.  The fp32 gather kernel these convs ran
// on before managed 27 % of the fp32 MFMA rate (0.25 ms per step for 0.65 % of the FLOPs).
constexpr int K1_SLAB_HALFS = 2 * 2 * F16X3_COT * 8;   // [part][kg][co][8] = 3072 halfs = 6 KB per (co tile, chunk)

__device__ __forceinline__ void f16x3_pack_k1_body(const float *__restrict__ w, _Float16 *__restrict__ out, const unsigned *hdr_in,
                                                   float *__restrict__ hdr_out, int Co, int Ci, int transposed, unsigned bid, unsigned nblk) {
    const float scale = weight_scale(hdr_in[2]);
    const int nchunks = Ci / F16X3_KC;
    const size_t n = (size_t)(Co / F16X3_COT) * nchunks * (K1_SLAB_HALFS / 2);
    for (size_t i = (size_t)bid * blockDim.x + threadIdx.x; i < n; i += (size_t)nblk * blockDim.x) {
        size_t r = i;
        const int e = (int)(r % 8); r /= 8;
        const int co = (int)(r % F16X3_COT); r /= F16X3_COT;
        const int kg = (int)(r % 2); r /= 2;
        const int chunk = (int)(r % nchunks);
        const int cot = (int)(r / nchunks);
        const int ci = chunk * F16X3_KC + kg * 8 + e, cog = cot * F16X3_COT + co;
        _Float16 hi, lo;
        // transposed: w is the original conv's [Ci][Co] weight, this pack its bwd-data conv
        split_f16(w[transposed ? (size_t)ci * Co + cog : (size_t)cog * Ci + ci] * scale, hi, lo);
        const size_t slab = (size_t)cot * nchunks + chunk;
        const size_t inner = ((size_t)kg * F16X3_COT + co) * 8 + e;
        out[slab * K1_SLAB_HALFS + inner] = hi;
        out[slab * K1_SLAB_HALFS + K1_SLAB_HALFS / 2 + inner] = lo;
    }
    if (bid == 0 && threadIdx.x == 0) {
        hdr_out[0] = 1.0f / scale;
        hdr_out[1] = scale;
    }
}
__global__ void __launch_bounds__(256) f16x3_pack_k1_kernel(const float *__restrict__ w, _Float16 *__restrict__ out, const unsigned *hdr_in,
                                                            float *__restrict__ hdr_out, int Co, int Ci, int transposed) {
    f16x3_pack_k1_body(w, out, hdr_in, hdr_out, Co, Ci, transposed, blockIdx.x, gridDim.x);
}

// KS = 1: four waves per workgroup, each with its own 64-voxel tile and the whole channel loop (large launches: the loop is
// covered by other waves).  KS > 1: the KS waves of a workgroup share ONE tile and split the channel chunks between them
// (contiguous ranges), then fold their accumulators through LDS in wave order — the small launches (<= one wave per SIMD on the
// chip) were a serial chain of nchunks dependent global-load round trips (12-48 x ~1.2 us: 28-41 us for 1-2 GFLOP).
// Either way the loads of chunk c+1 are issued before the MFMAs of chunk c (two register sets).
template <int KS, int NTT = 2>
__global__ void __launch_bounds__(KS == 1 ? 256 : 64 * KS) __attribute__((amdgpu_waves_per_eu(2)))
conv3d_k1_f16x3_kernel(const float *__restrict__ x, const _Float16 *__restrict__ wslabs, const float *__restrict__ whdr,
                       const float *__restrict__ bias, float *__restrict__ y, int N, int Ci, int Co, int DHW, unsigned x_bytes,
                       const float *__restrict__ x_range) {
    constexpr int MT = 3, NT = NTT;   // NT: 32-voxel column tiles per wave
    float x_scale = X_SCALE, x_unscale = 1.0f / X_SCALE;
    if (x_range) range_scale_block(x_range, x_scale, x_unscale);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, kg = lane >> 5;
    const long v0 = (KS == 1 ? (long)blockIdx.x * 4 + wave : (long)blockIdx.x) * (NT * 32);   // this wave's 64 voxels (DHW % 64 == 0: one sample)
    if (v0 >= (long)N * DHW) return;                                 // wave-uniform (KS > 1: workgroup-uniform)
    const int n = (int)(v0 / DHW), r0 = (int)(v0 - (long)n * DHW);
    const int cot = blockIdx.y, nchunks = Ci / F16X3_KC;
    const int cps = (nchunks + KS - 1) / KS;
    const int c_begin = KS == 1 ? 0 : wave * cps, c_end = KS == 1 ? nchunks : min(nchunks, c_begin + cps);
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)x_bytes, 0x00020000);
    const unsigned vbase = (unsigned)((((long)n * Ci + kg * 8) * DHW + r0 + j) * 4);   // (n, ci = kg*8, voxel j of tile 0)
    const unsigned cstride = (unsigned)DHW * 4u;
    const _Float16 *wl = wslabs + (size_t)cot * nchunks * K1_SLAB_HALFS + (kg * F16X3_COT + j) * 8;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[m][t][q] = 0.0f;

    unsigned sat_ = 0;
    float xv[2][NT][8];
    half8 ah[2][MT], al[2][MT];
#define K1_LOAD(set, c_)                                                                                              \
    {                                                                                                                 \
        const unsigned soff_ = (unsigned)((long)(c_) * F16X3_KC * DHW * 4);                                           \
        _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                                \
            _Pragma("unroll") for (int e = 0; e < 8; ++e)                                                             \
                xv[set][t][e] = buf_load_f(rsrc, vbase + (unsigned)t * 128u + (unsigned)e * cstride, soff_);          \
        const _Float16 *ws_ = wl + (size_t)(c_) * K1_SLAB_HALFS;                                                      \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                              \
            ah[set][m] = *reinterpret_cast<const half8 *>(ws_ + m * 32 * 8);                                          \
            al[set][m] = *reinterpret_cast<const half8 *>(ws_ + K1_SLAB_HALFS / 2 + m * 32 * 8);                      \
        }                                                                                                             \
    }
#define K1_MFMA(set)                                                                                                  \
    {                                                                                                                 \
        half8 bh[NT], bl[NT];                                                                                         \
        _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                                \
            _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                           \
                _Float16 h, l;                                                                                        \
                const float sv = xv[set][t][e] * x_scale;                                                             \
                sat_ += !(fabsf(sv) <= F16_CLAMP);                                                                    \
                split_f16(sv, h, l);                                                                                  \
                bh[t][e] = h;                                                                                         \
                bl[t][e] = l;                                                                                         \
            }                                                                                                         \
        _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                                \
            _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                            \
                acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[set][m], bh[t], acc[m][t], 0, 0, 0);            \
        _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                                \
            _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                            \
                acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[set][m], bl[t], acc[m][t], 0, 0, 0);            \
        _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                                \
            _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                            \
                acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[set][m], bh[t], acc[m][t], 0, 0, 0);            \
    }
    if (c_begin < c_end) K1_LOAD(0, c_begin)
    for (int c = c_begin; c < c_end; c += 2) {
        if (c + 1 < c_end) K1_LOAD(1, c + 1)
        K1_MFMA(0)
        if (c + 1 < c_end) {
            if (c + 2 < c_end) K1_LOAD(0, c + 2)
            K1_MFMA(1)
        }
    }
#undef K1_LOAD
#undef K1_MFMA
    const float unscale = whdr[0] * x_unscale;
    const int co0 = cot * F16X3_COT;
    if (KS == 1) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int co = co0 + m * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kg;
                const float bv = bias ? bias[co] : 0.0f;
                float *dv = y + ((size_t)n * Co + co) * DHW + r0 + j;
#pragma unroll
                for (int t = 0; t < NT; ++t) dv[t * 32] = acc[m][t][reg] * unscale + bv;
            }
        }
    } else {
        // fold the KS partial accumulators: per row tile m every wave parks its 32 registers in LDS; wave w then sums (in wave
        // order: deterministic) and stores the registers w, w + KS, ...
        __shared__ float red[KS > 1 ? KS : 1][NT * 16 * 64];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) red[wave][(t * 16 + reg) * 64 + lane] = acc[m][t][reg];
            __syncthreads();
            for (int q = wave; q < NT * 16; q += KS) {
                const int t = q >> 4, reg = q & 15;
                float sum = red[0][q * 64 + lane];
#pragma unroll
                for (int k = 1; k < KS; ++k) sum += red[k][q * 64 + lane];
                const int co = co0 + m * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kg;
                const float bv = bias ? bias[co] : 0.0f;
                y[((size_t)n * Co + co) * DHW + r0 + j + t * 32] = sum * unscale + bv;
            }
            __syncthreads();
        }
    }
    if (__builtin_amdgcn_ballot_w64(sat_ != 0) != 0) {  // never taken in normal operation (non-finite operands)
        unsigned tot = sat_;
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) tot += __shfl_xor(tot, sft, 64);
        if (lane == 0) atomicAdd(&g_f16x3_saturated, (unsigned long long)tot);
    }
}

// waves sharing a tile (1 = none): small launches split the channel loop 4 or 8 ways
static int f16x3_k1_ksplit(int N, int Ci, int Co, int DHW) {
    const char *force = getenv("MPHIP_F16X3_K1_KS");   // dev: same-box A/B
    if (force && (atoi(force) == 1 || atoi(force) == 4 || atoi(force) == 8)) return atoi(force);
    const long wave_tiles = (long)N * DHW / 64 * (Co / F16X3_COT);
    const int nchunks = Ci / F16X3_KC;
    if (wave_tiles >= 1024 || nchunks < 8) return 1;   // (measured at B=8: 192->96 @8x32x32, 1024 wave tiles: 26 us unsplit, 31 us split 4)
    return (wave_tiles <= 512 && nchunks >= 16) ? 8 : 4;
}

static long f16x3_k1_min_voxels() {
    const char *e = getenv("MPHIP_F16X3_K1_MIN");   // dev: threshold A/B
    return e ? atol(e) : 1024;
}

bool f16x3_supported(int N, int Ci, int Co, int D, int H, int W, int k) {
    if (k == 1)   // the k=1 GEMM kernel: whole 64-voxel wave tiles inside one sample, and enough of them to beat the split-K
                  // fp32 gather kernel (measured: 1024 voxels 29 vs 32 us at B=8, but slower below — B=1 went 1.93 -> 2.11 ms)
        return Ci % F16X3_KC == 0 && Co % F16X3_COT == 0 && ((long)D * H * W) % 64 == 0 && (long)N * D * H * W >= f16x3_k1_min_voxels() &&
               (size_t)N * Ci * D * H * W * 4 < 0x80000000ull;
    return k == 3 && Ci % F16X3_KC == 0 && Co % F16X3_COT == 0 && H % 8 == 0 && W % 8 == 0 && D % 2 == 0 &&
           (size_t)N * Ci * D * H * W * 4 < 0x80000000ull;
}

size_t f16x3_packed_bytes_k1(int Co, int Ci) {
    return 16 + (size_t)(Co / F16X3_COT) * (Ci / F16X3_KC) * K1_SLAB_HALFS * sizeof(_Float16);
}

static size_t f16x3_direct_bytes(int Co, int Ci) {   // header + the direct kernel's slabs
    return 16 + (size_t)(Co / F16X3_COT) * (Ci / F16X3_KC) * F16X3_NG * SLAB_HALFS * sizeof(_Float16);
}

size_t f16x3_packed_bytes(int Co, int Ci) {   // ... + the transformed-domain slabs of the layers that can take that kernel
    return f16x3_direct_bytes(Co, Ci) + f16x3_wino_packed_bytes(Co, Ci);
}

F16x3Plan f16x3_plan(int N, int Ci, int Co, int D, int H, int W, bool roi) {
    F16x3Plan p;
    p.td = D % 4 == 0 ? 4 : 2;
    // variant: 0 = the direct kernels ((td,8,8) tile, 256 / 128 voxels per workgroup); 4 = the F(2,3) kernel.  (Until r05 a variant 1 —
    // a (4,8,16) tile, 512 voxels per workgroup, the r02-r03 kernel of the chip-filling launches — was kept as the fallback of
    // MPHIP_WINOGRAD=0; it was the one hot instantiation with scratch (248-256 B) and every launch it could take is variant 4's: removed.)
    const char *force = getenv("MPHIP_F16X3_TILE");   // dev knob, read per call: "0" = direct kernels only (tools/sweep_conv_plans.py)
    const int cot = Co / F16X3_COT;
    p.variant = 0;
    (void)roi;
    // variant 4: the 1-D Winograd F(2,3) kernel (conv3d_f16x3_wino.hip; (4,8,8) tile, 2/3 of the MFMAs) on launches that fill the chip
    // (demand-driven launches follow the full launch's choice, so that the tiles they compute carry the same bits)
    // (depth-2 volumes: the F(2,3) kernel's two-frame mode — not for demand-driven launches)
    if (!force && !(roi && D == 2) && f16x3_wino_usable(N, Ci, Co, D, H, W)) p.variant = 4;
    const long tiles = p.variant == 4 ? f16x3_wino_tiles(N, D, H, W) : (long)N * (D / p.td) * (H / 8) * (W / 8);
    const int nchunks = Ci / F16X3_KC;
    // split-K only when the launch cannot give every CU a workgroup (each split adds a slab write + a reduce pass): the largest
    // whole-chunk split that still fits the chip in ONE round of resident workgroups (one per CU; two for the 4-wave (2,8,8) kernel).
    // r03 sweep (tools/sweep_conv_plans.py): a second round costs more than it hides (B=8, 384->192 @4x16x16: 256 workgroups 98 us,
    // 512 106 us), and below that one chunk per workgroup beats three (B=1, 768->384 @2x8x8: 16 splits 41 us, 48 splits 24 us — the
    // launch is one workgroup's serial chain of chunks).
    int sp = 1;
    {
        const long base = tiles * cot, slots = (p.variant == 0 && p.td == 2) ? 512 : 256;
        static const char *old_rule = getenv("MPHIP_F16X3_OLD_SPLITS");   // dev: same-box A/B against the r02 rule
        if (old_rule) {
            if (base < 256)
                while (base * sp < 512 && nchunks / (sp * 2) >= 3) sp *= 2;
        } else if (base < slots) {
            for (int dv = 2; dv <= nchunks; ++dv)
                if (nchunks % dv == 0 && base * dv <= slots) sp = dv;
        }
    }
    const char *force_sp = getenv("MPHIP_F16X3_SPLITS");   // dev: planner sweep (tools/sweep_conv_plans.py)
    if (force_sp && atoi(force_sp) > 0 && nchunks % atoi(force_sp) == 0) sp = atoi(force_sp);   // (whole chunks per split only)
    if (p.variant == 4) sp = f16x3_wino_splits(N, Ci, Co, D, H, W);
    p.splits = sp;
    p.chunks_per_split = (nchunks + sp - 1) / sp;
    p.grid = dim3((unsigned)tiles, Co / F16X3_COT, sp);
    return p;
}

int f16x3_tile_waves(const F16x3Plan &p) {   // GroupNorm-partial rows per tile of the kernel variant f16x3_launch picks (= its waves;
                                              // the Winograd kernel leaves one row per plane pair)
    return p.variant == 4 ? 2 : p.td == 4 ? 8 : 4;
}

void f16x3_tile_dims(const F16x3Plan &p, int dims[3]) {   // output tile (d,h,w) of the kernel variant f16x3_launch picks
    dims[0] = p.variant == 4 ? 4 : p.td;
    dims[1] = 8;
    dims[2] = 8;
}

int f16x3_pack(const float *w, void *out, int Co, int Ci, int k, int transposed, const void *header_from, hipStream_t s) {
    const unsigned *hdr = (const unsigned *)out;
    if (header_from) {
        hdr = (const unsigned *)header_from;  // max|w| already known (the same weight tensor packed for the other direction)
    } else {
        zero_fill(out, 16, s);  // header: the absmax kernel accumulates with atomicMax
        const size_t n = (size_t)Co * Ci * (k == 3 ? 27 : 1);
        hipLaunchKernelGGL(f16x3_absmax_kernel, dim3((unsigned)std::min<size_t>(2048, (n + 4095) / 4096)), dim3(256), 0, s, w, n,
                           (unsigned *)out);
    }
    if (k == 1)
        hipLaunchKernelGGL(f16x3_pack_k1_kernel, dim3(256), dim3(256), 0, s, w, (_Float16 *)((char *)out + 16), hdr, (float *)out, Co,
                           Ci, transposed);
    else {
        hipLaunchKernelGGL(f16x3_pack_kernel, dim3((unsigned)((Co / PK_CO) * (Ci / F16X3_KC))), dim3(256), 0, s, w,
                           (_Float16 *)((char *)out + 16), hdr, (float *)out, Co, Ci, transposed,
                           f16x3_wino_packed_bytes(Co, Ci) ? (_Float16 *)((char *)out + f16x3_direct_bytes(Co, Ci)) : (_Float16 *)nullptr);
    }
    return check_launch("pack_conv_weight(f16x3)");
}

// ---- every weight of a module in one launch per kernel kind (training re-packs all of them every step: r05's step spent 1.08 ms in
// 105 latency-bound pack launches).  A block finds its job by its index in the launch (binary search over the jobs' first blocks) and runs
// the body of the single-weight kernel on it: same bits as the single calls.
__device__ __forceinline__ int pack_find(const int *__restrict__ first, int n, int bid) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (first[mid] <= bid) lo = mid; else hi = mid - 1;
    }
    return lo;
}
__device__ __forceinline__ const unsigned *pack_hdr_in(const PackJob &j) { return (const unsigned *)(j.like ? j.like : j.wp); }
__global__ void __launch_bounds__(256) pack_many_zero_hdr_kernel(const PackJob *__restrict__ jobs, const int *__restrict__ sel, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n * 4) reinterpret_cast<unsigned *>(jobs[sel[i >> 2]].wp)[i & 3] = 0u;
}
__global__ void __launch_bounds__(256) pack_many_absmax_kernel(const PackJob *__restrict__ jobs, const int *__restrict__ sel,
                                                               const int *__restrict__ first, int n) {
    const int q = pack_find(first, n, (int)blockIdx.x);
    const PackJob j = jobs[sel[q]];
    f16x3_absmax_body(j.w, (size_t)j.Co * j.Ci * (j.k == 3 ? 27 : 1), (unsigned *)j.wp, blockIdx.x - first[q], first[q + 1] - first[q]);
}
__global__ void __launch_bounds__(256) pack_many_k3_kernel(const PackJob *__restrict__ jobs, const int *__restrict__ sel,
                                                           const int *__restrict__ first, int n) {
    __shared__ __attribute__((aligned(16))) float tile[PK_LDS_FLOATS];
    const int q = pack_find(first, n, (int)blockIdx.x);
    const PackJob j = jobs[sel[q]];
    char *out = (char *)j.wp;
    f16x3_pack_body(j.w, (_Float16 *)(out + 16), pack_hdr_in(j), (float *)out, j.Co, j.Ci, j.transposed,
                    j.wino_off ? (_Float16 *)(out + j.wino_off) : (_Float16 *)nullptr,
                    (int)blockIdx.x - first[q], tile);
}
__global__ void __launch_bounds__(256) pack_many_k1_kernel(const PackJob *__restrict__ jobs, const int *__restrict__ sel,
                                                           const int *__restrict__ first, int n) {
    const int q = pack_find(first, n, (int)blockIdx.x);
    const PackJob j = jobs[sel[q]];
    char *out = (char *)j.wp;
    f16x3_pack_k1_body(j.w, (_Float16 *)(out + 16), pack_hdr_in(j), (float *)out, j.Co, j.Ci, j.transposed, blockIdx.x - first[q],
                       first[q + 1] - first[q]);
}

size_t f16x3_pack_wino_offset(int Co, int Ci) { return f16x3_wino_packed_bytes(Co, Ci) ? f16x3_direct_bytes(Co, Ci) : 0; }

int f16x3_pack_blocks(const PackJob &j, int kind) {   // blocks of job j in the absmax (0) / k=3 (1) / k=1 (2) launch of f16x3_pack_many
    const size_t n = (size_t)j.Co * j.Ci * (j.k == 3 ? 27 : 1);
    if (kind == 0) return (int)std::min<size_t>(2048, (n + 4095) / 4096);
    if (kind == 1) return (j.Co / PK_CO) * (j.Ci / F16X3_KC);
    return (int)std::min<size_t>(256, (n + 255) / 256);
}

int f16x3_pack_many(const PackJob *jobs, PackSel absmax, PackSel k3, PackSel k1, hipStream_t s) {
    if (absmax.n) {
        hipLaunchKernelGGL(pack_many_zero_hdr_kernel, dim3((unsigned)((absmax.n * 4 + 255) / 256)), dim3(256), 0, s, jobs, absmax.job, absmax.n);
        hipLaunchKernelGGL(pack_many_absmax_kernel, dim3((unsigned)absmax.blocks), dim3(256), 0, s, jobs, absmax.job, absmax.first, absmax.n);
    }
    if (k3.n) hipLaunchKernelGGL(pack_many_k3_kernel, dim3((unsigned)k3.blocks), dim3(256), 0, s, jobs, k3.job, k3.first, k3.n);
    if (k1.n) hipLaunchKernelGGL(pack_many_k1_kernel, dim3((unsigned)k1.blocks), dim3(256), 0, s, jobs, k1.job, k1.first, k1.n);
    return check_launch("pack_conv_weights(f16x3)");
}

int f16x3_launch_k1(const float *x, const void *wpacked, const float *bias, float *dst, int N, int Ci, int Co, int DHW,
                    const float *x_range, hipStream_t s) {
    const float *hdr = (const float *)wpacked;
    const _Float16 *slabs = (const _Float16 *)((const char *)wpacked + 16);
    const unsigned xb = (unsigned)((size_t)N * Ci * DHW * 4);
    const long waves = (long)N * DHW / 64;
    const int ks = f16x3_k1_ksplit(N, Ci, Co, DHW);
    // a launch with ONE wave per SIMD (64-voxel wave tiles) runs 32-voxel tiles on twice the waves (192->96 @8x32x32, B=8: 29 -> 25 us; with two
    // waves per SIMD already — 96->192 — it is slower, 26 -> 31 us: the weight fragments are re-read per wave)
    static const char *nt_env = getenv("MPHIP_F16X3_K1_NT");   // dev: same-box A/B (1 / 2 forces)
    const bool nt1 = ks == 1 && (nt_env ? atoi(nt_env) == 1 : waves * (Co / F16X3_COT) <= 1024);
    if (ks == 8)
        hipLaunchKernelGGL(conv3d_k1_f16x3_kernel<8>, dim3((unsigned)waves, Co / F16X3_COT), dim3(512), 0, s, x, slabs, hdr, bias, dst, N, Ci,
                           Co, DHW, xb, x_range);
    else if (ks == 4)
        hipLaunchKernelGGL(conv3d_k1_f16x3_kernel<4>, dim3((unsigned)waves, Co / F16X3_COT), dim3(256), 0, s, x, slabs, hdr, bias, dst, N, Ci,
                           Co, DHW, xb, x_range);
    else if (nt1)   // 32 voxels per wave: twice the waves of a launch that has one or two per SIMD
        hipLaunchKernelGGL((conv3d_k1_f16x3_kernel<1, 1>), dim3((unsigned)((2 * waves + 3) / 4), Co / F16X3_COT), dim3(256), 0, s, x, slabs, hdr,
                           bias, dst, N, Ci, Co, DHW, xb, x_range);
    else
        hipLaunchKernelGGL(conv3d_k1_f16x3_kernel<1>, dim3((unsigned)((waves + 3) / 4), Co / F16X3_COT), dim3(256), 0, s, x, slabs, hdr, bias,
                           dst, N, Ci, Co, DHW, xb, x_range);
    return check_launch("conv3d_fwd(f16x3, k=1)");
}

int f16x3_launch(const F16x3Plan &p, const float *x, const void *wpacked, const float *bias, float *dst, int N, int Ci,
                 int Co, int D, int H, int W, const float *in_affine, int in_relu, const float *x_scale, hipStream_t s, const int *roi,
                 int roi_frames, int *tile_list, int roi_dilate, float *gn_part, hipEvent_t t0, hipEvent_t t1) {
    if (in_affine && Ci > 768) {
        set_error("conv3d_fwd(f16x3): fused input GroupNorm supports Ci <= 768 (got %d)", Ci);
        return MPHIP_EINVAL;
    }
    const float *hdr = (const float *)wpacked;
    const _Float16 *slabs = (const _Float16 *)((const char *)wpacked + 16);
    const unsigned xb = (unsigned)((size_t)N * Ci * D * H * W * 4);
    if (roi) {   // demand-driven: boxes -> the list of tiles they touch (1 + tiles ints of caller workspace); the kernel gets the LIST
        int dims[3];
        f16x3_tile_dims(p, dims);
        // (r03: extra workgroups of this launch pre-reading the packed weights into every XCD's L2 changed nothing — a demand-driven
        //  tile is one CU's MFMA work, 2592 K-steps x 9 MFMAs per wave, not a chain of L2 misses)
        hipLaunchKernelGGL(roi_tile_list_kernel, dim3(1), dim3(1024), 0, s, roi, roi_frames, (int)p.grid.x, D, H, W, dims[0], dims[1], dims[2],
                           roi_dilate, tile_list);
        roi = tile_list;
    }
    if (p.variant == 4) {   // the 1-D Winograd F(2,3) kernel (full launches and demand-driven ones alike: a listed tile carries the full launch's bits)
        if (in_affine && Ci > 384 && D != 2) {
            set_error("conv3d_fwd(f16x3, F(2,3)): fused input GroupNorm supports Ci <= 384 (got %d)", Ci);
            return MPHIP_EINVAL;
        }
        return f16x3_wino_launch(x, (const char *)wpacked + f16x3_direct_bytes(Co, Ci), hdr, bias, dst, N, Ci, Co, D, H, W, p.splits,
                                 in_affine, in_relu, x_scale, s, roi, gn_part, t0, t1);
    }
    // persistent grid: as many workgroups as the chip runs at once (LDS: one per CU for the two big variants, two for
    // the (2,8,8) one), each walking its share of the tiles
    const int tiles_total = (int)p.grid.x;
    const int per_cu = p.td == 4 ? 1 : 2;
    static const bool thirds_off = getenv("MPHIP_ROI_THIRDS") && getenv("MPHIP_ROI_THIRDS")[0] == '0';   // dev: same-box A/B
    const bool thirds = roi && p.variant == 0 && p.td == 4 && !gn_part && !thirds_off;
    const long others = (long)p.grid.y * (thirds ? 3 : 1) * p.grid.z;
    static const char *cus_s = getenv("MPHIP_CONV_CUS");   // dev: persistent grid size (leave CUs to another batch's small kernels)
    const long cus = cus_s ? atol(cus_s) : 256;
    long gx = (cus * per_cu + others - 1) / others;
    if (gx < 1) gx = 1;
    if (gx > tiles_total || getenv("MPHIP_F16X3_NO_PERSIST")) gx = tiles_total;
    dim3 grid((unsigned)gx, p.grid.y * (thirds ? 3 : 1), p.grid.z);
    static const int xcd_on = !(getenv("MPHIP_F16X3_XCD") && getenv("MPHIP_F16X3_XCD")[0] == '0');  // dev switch for same-box A/B
    // (two-slab groups for the 512-voxel tile — 5 instead of 9 barriers per chunk, 147 KB of LDS — were tried: the
    //  compiler spills 188 registers in that instantiation and it runs 35 % slower)
    // (t0, t1: the measurement hook's events ride on the kernel command itself — hipExtLaunchKernelGGL stamps them with the kernel's own
    //  begin / end, so what another stream's kernel makes this launch WAIT for CUs is not counted as its duration)
#define F16X3_LAUNCH(kern_, block_)                                                                                              \
    {                                                                                                                            \
        if (t0 && t1)                                                                                                            \
            hipExtLaunchKernelGGL(kern_, grid, dim3(block_), 0, s, t0, t1, 0, x, slabs, hdr, bias, dst, N, Ci, Co, D, H, W,      \
                                  p.chunks_per_split, xb, in_affine, in_relu, x_scale, tiles_total, xcd_on, roi, roi_frames, gn_part); \
        else                                                                                                                     \
            hipLaunchKernelGGL(kern_, grid, dim3(block_), 0, s, x, slabs, hdr, bias, dst, N, Ci, Co, D, H, W, p.chunks_per_split, xb, \
                               in_affine, in_relu, x_scale, tiles_total, xcd_on, roi, roi_frames, gn_part);                      \
    }
    if (p.td == 4 && thirds) F16X3_LAUNCH((conv3d_k3_f16x3_third_kernel<4, 8, 8, 8, 3>), 512)
    else if (p.td == 4) F16X3_LAUNCH((conv3d_k3_f16x3_kernel<4, 8, 8, 8, 3>), 512)
    else F16X3_LAUNCH((conv3d_k3_f16x3_kernel<2, 8, 8, 4, 1>), 256)
#undef F16X3_LAUNCH
    return check_launch("conv3d_fwd(f16x3)");
}

}  // namespace mphip

#ifdef MPHIP_PROFILE_PHASES
extern "C" int mphip_debug_f16x3_profile(unsigned long long *out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(mphip::g_f16x3_prof), 64) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(mphip::g_f16x3_prof), z, 64) != hipSuccess) return -1;
    }
    return 0;
}
#endif

extern "C" int mphip_f16x3_saturation_count(unsigned long long *count, int reset) {
    // synchronous (copies from the device): a diagnostic, not part of the stream-ordered path
    MPHIP_REQUIRE(count, "f16x3_saturation_count: null pointer");
    if (hipMemcpyFromSymbol(count, HIP_SYMBOL(mphip::g_f16x3_saturated), sizeof(unsigned long long)) != hipSuccess) {
        mphip::set_error("f16x3_saturation_count: hipMemcpyFromSymbol failed");
        return MPHIP_ELAUNCH;
    }
    if (reset) {
        const unsigned long long z = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(mphip::g_f16x3_saturated), &z, sizeof(z)) != hipSuccess) {
            mphip::set_error("f16x3_saturation_count: hipMemcpyToSymbol failed");
            return MPHIP_ELAUNCH;
        }
    }
    unsigned long long wn = 0;   // the transformed-domain kernel keeps its own counter (separate translation unit)
    if (mphip::f16x3_wino_saturation(&wn, reset) != 0) {
        mphip::set_error("f16x3_saturation_count: hipMemcpyFromSymbol failed");
        return MPHIP_ELAUNCH;
    }
    *count += wn;
    return MPHIP_OK;
}
