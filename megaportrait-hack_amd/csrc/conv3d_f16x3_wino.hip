// K4 fast mode, transformed domain — Conv3d 3x3x3 on the f16 matrix cores ("f16x3" split arithmetic) with a 1-D Winograd
// F(2,3) transform along W: two thirds of the matrix work of conv3d_f16x3.hip for the same fp32-class result.
//
// Why: the direct f16x3 kernel is bound by the energy of its MFMAs (the package sits at its power limit while it runs:
// DESIGN.md 3, profiles/r02_power_probe.txt; 63 % of a launch's energy is the 23.9 M v_mfma_f32_32x32x16_f16 it issues).  Only
// fewer MFMAs per output voxel make it faster.  F(2,3) along w computes an output PAIR (w = 2q, 2q+1) of one (d,h) row from
// four input values x[2q-1 .. 2q+2] per (input channel, kd, kh) with 4 multiplies instead of 6:
//     t0 = x0 - x2   t1 = x1 + x2   t2 = x2 - x1   t3 = x1 - x3                      (input transform, fp32, before the split)
//     u0 = g0        u1 = (g0+g1+g2)/2   u2 = (g0-g1+g2)/2   u3 = g2                 (filter transform, at pack time)
//     M_p[co][q] = sum over (ci, kd, kh) of u_p * t_p                                  (four independent GEMMs, K = 9 Ci)
//     y[2q] = M0 + M1 + M2     y[2q+1] = M1 - M2 - M3                                   (output transform, fp32)
// Each u_p / t_p is split into two f16 halves like the direct kernel's operands (three MFMAs per product, fp32 accumulate):
// |t| <= 2 max|x| and |u| <= 1.5 max|w| stay inside the f16 range at the same per-tensor power-of-two scales.
//
// Shape of the kernel (512 threads = 8 waves, two per SIMD, one workgroup per CU, persistent over tiles):
//   * tile = 4 x 8 x 8 output voxels x 96 output channels = 4 positions x 128 pair-columns; wave (p, ch) owns position p of the
//     two d-planes 2ch, 2ch+1: 3 x 2 MFMA tiles, 96 accumulator registers — the direct kernel's wave shape, so the LDS reads
//     per MFMA are the same.  (Accumulators per output voxel double in the transformed domain; that is why the tile is 256
//     voxels, not 512.)
//   * X: the halo rows (6 x 10 rows of 10 voxels per channel) are loaded one (channel pair, row) per thread, transformed and split
//     in registers, and written to LDS as [part][position][k-group][row][pair][8 channels] — the pair domain has no halo in w,
//     so the 32 columns of an MFMA tile (8 rows x 4 pairs of one plane) are 512 contiguous bytes per k-group: conflict-free
//     ds_read_b128 without any lane permutation.
//   * W: one slab = one (kd,kh) x 16 input channels x 4 positions x 96 output channels = 24 KB, streamed by LDS-DMA into a
//     RING of four slabs, three slabs ahead of the MFMAs.  The DMA is issued by hand (mphip_f16x3.h: lds_dma16) so that no
//     compiler-inserted vmcnt(0) sits between a transfer and the fragment reads of OTHER slabs, and a wave waits only for the
//     pieces it issued two intervals ago.  A-fragments of the next slab are prefetched BEFORE the interval's barrier (the slab
//     was published one barrier earlier), so the barrier only guards the ring slot that is overwritten next.
//   * output transform: the four positions of a pair live in four waves.  After the last chunk every wave parks three quarters
//     of its accumulators in the (now dead) X region, one 32-channel row tile per round, and finishes the quarter it keeps:
//     y0/y1, bias, the GroupNorm statistics partials of its channels, 8-byte stores.
#include <stdlib.h>

#include <hip/hip_ext.h>

#include "mphip_ablate.h"
#include "mphip_conv.h"
#include "mphip_f16x3.h"

namespace mphip {

constexpr int WN_KC = 16;                                  // input channels per chunk = K of one MFMA
constexpr int WN_COT = 96;                                 // output channels per workgroup (3 MFMA row tiles)
constexpr int WN_NG = 9;                                   // (kd,kh) slabs per chunk
constexpr int WN_RING = 4;                                 // slabs resident in LDS
constexpr int WN_SLAB_HALFS = 2 * 4 * 2 * WN_COT * 8;      // [part][position][kg][co][8] = 12288 halfs = 24576 B
constexpr int WN_PART_HALFS = WN_SLAB_HALFS / 2;
constexpr int WN_TD = 4, WN_TH = 8, WN_TW = 8;             // output tile
constexpr int WN_HD = WN_TD + 2, WN_HH = WN_TH + 2;
constexpr int WN_ROWS = WN_HD * WN_HH;                     // 60 halo rows
constexpr int WN_XBLK = WN_ROWS * 4 * 8 + 16;              // halfs per (part, position, kg) block: 240 (row, pair) slots x 8 channels
                                                           // + 32 B so that the two k-groups of a staging write sit 8 banks apart
constexpr int WN_XPART = 4 * 2 * WN_XBLK;                  // halfs per part (hi or lo)
constexpr int WN_X_HALFS = 2 * WN_XPART;                   // 30976 halfs = 61952 B
constexpr int WN_AFF_CI = 384;                             // fused input GroupNorm table: Ci <= 384 (LDS: 98304 + 61952 + 3072 B + the range fold's 68)
constexpr int WN_XLOADS = 8;                               // vector-memory instructions of one halo prefetch (per wave)

// Weight layout (written by f16x3_pack_kernel in conv3d_f16x3.hip, from the tile it stages for the direct slabs):
//   slabs[(cot*nchunks + chunk)*9 + (kd*3+kh)][part][position][kg][co][8] f16, behind the direct slabs of the same pack;
//   scale: the per-tensor power of two of the pack's header, max|w|*scale < 2^15, so |u|*scale < 1.5 * 2^15 < 65504.

__device__ unsigned long long g_f16x3_wino_saturated;

#ifdef MPHIP_WN_TRACE   /* dev: wall-clock (100 MHz) stamps per workgroup — start, end of every tile (up to 14) — tools/dbg_wino_trace.py */
__device__ unsigned long long g_wn_trace[1024 * 16];
#define WN_STAMP(i) if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 1024 && (i) < 16) g_wn_trace[blockIdx.x * 16 + (i)] = wall_clock64();
#else
#define WN_STAMP(i)
#endif
#ifdef MPHIP_PROFILE_PHASES
// dev instrumentation (itself intrusive, ~+10 % wave cycles): shader cycles per phase, summed over all waves:
// [0] prologue  [1] interval: DMA issue + fragment reads + MFMA issue  [2] interval: wait for this wave's DMA pieces  [3] interval: barrier
// [4] halo write (transform + split + LDS stores)  [5] its barrier + first fragment reload  [6] output transform + epilogue  [7] waves
__device__ unsigned long long g_f16x3_wino_prof[8];
#define WPROF_DECL unsigned long long pt_ = __builtin_readcyclecounter(), pa_[7] = {0, 0, 0, 0, 0, 0, 0}
#define WPROF_ADD(slot) { const unsigned long long n_ = __builtin_readcyclecounter(); pa_[slot] += n_ - pt_; pt_ = n_; }
#define WPROF_FLUSH if ((threadIdx.x & 63) == 0) { for (int q_ = 0; q_ < 7; ++q_) atomicAdd(&g_f16x3_wino_prof[q_], pa_[q_]); atomicAdd(&g_f16x3_wino_prof[7], 1ull); }
#else
#define WPROF_DECL
#define WPROF_ADD(slot)
#define WPROF_FLUSH
#endif

__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
conv3d_k3_f16x3_wino_kernel(const float *__restrict__ x, const _Float16 *__restrict__ wslabs, const float *__restrict__ whdr,
                            const float *__restrict__ bias, float *__restrict__ y, int N, int Ci, int Co, int D, int H, int W,
                            int chunks_per_split, unsigned x_bytes, const float *__restrict__ in_affine, int in_relu,
                            const float *__restrict__ x_range, int tiles_total, int xcd_aware, const int *__restrict__ tile_list,
                            float *__restrict__ gn_part) {
    __shared__ __attribute__((aligned(16))) _Float16 smem[WN_RING * WN_SLAB_HALFS + WN_X_HALFS + WN_AFF_CI * 4];
    _Float16 *const Ws = smem;                                   // ring of 4 slabs
    _Float16 *const Xs = smem + WN_RING * WN_SLAB_HALFS;         // [part][position][kg][row*4 + pair][8]
    float *const aff = reinterpret_cast<float *>(Xs + WN_X_HALFS);   // [Ci][2]: (scale, shift) of the fused input GroupNorm
    float *const Ex = reinterpret_cast<float *>(Xs);             // output-transform exchange: [wave][slot 0..5][lane][4] (48 KB)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = wave & 3, ch = wave >> 2;                      // Winograd position, plane pair
    const int j = lane & 31, kgl = lane >> 5;
    const int HW = H * W, DHW = D * HW;
    // Demand-driven evaluation (tile_list = {count, id, id, ...}, roi_tile_list_kernel): only the listed output tiles are computed, dealt
    // round-robin over the persistent workgroups — same arithmetic per tile, so a listed tile carries the full launch's bits.
    const int ntiles = tile_list ? tile_list[0] : tiles_total;
    auto tile_at = [&](int jj) -> int { return tile_list ? tile_list[1 + jj] : jj; };
    const int j_first = (tile_list || !xcd_aware) ? (int)blockIdx.x : (int)xcd_remap(blockIdx.x, gridDim.x);
    if (j_first >= ntiles) return;   // (workgroup-uniform, before any barrier)
    float x_scale = 16.0f, x_unscale = 1.0f / 16.0f;
    if (x_range) range_scale_block(x_range, x_scale, x_unscale);

    const int tiles_w = W / WN_TW, tiles_h = H / WN_TH, tiles_d = D / WN_TD;
    int n, d0, h0, w0;   // the tile the STAGING side addresses (one chunk ahead of the MFMAs; the next tile during a tile's last chunk)
    auto decode_tile = [&](int tile_id) {
        int bid = tile_id;
        const int tw = bid % tiles_w; bid /= tiles_w;
        const int th = bid % tiles_h; bid /= tiles_h;
        const int td = bid % tiles_d;
        n = bid / tiles_d;
        d0 = td * WN_TD; h0 = th * WN_TH; w0 = tw * WN_TW;
    };
    decode_tile(tile_at(j_first));
    const int cot = blockIdx.y;
    const int nchunks = Ci / WN_KC;
    // split-K (blockIdx.z): launches that cannot give every CU a tile split the input channels in whole chunks; the output transform is
    // linear, so every split transforms its own partial sums and writes them to its slab [z][N,Co,DHW] (bias added by the ordered reduce)
    const int c_begin = blockIdx.z * chunks_per_split, c_end = min(nchunks, c_begin + chunks_per_split);
    if (c_begin >= c_end) return;
    const int nmine = (ntiles - j_first + (int)gridDim.x - 1) / (int)gridDim.x;   // tiles this workgroup walks
    const int s_total = nmine * (c_end - c_begin) * WN_NG;                         // slabs it consumes

    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)x_bytes, 0x00020000);
    const unsigned chan_stride = (unsigned)DHW * 4u;

    // ---- X staging: thread = (channel pair cp, halo row srow); 480 of the 512 threads ------------------------------------------
    // (threads 480..511 re-do row 59 — same loads, same values, same LDS addresses — so that the staging code is branch-free: it is
    //  issued into the shadow of a chunk's last MFMAs, and the scheduler interleaves only inside one basic block)
    const int cp = tid & 7, srow = min(tid >> 3, WN_ROWS - 1);
    const bool stager = true;
    const int sdl = srow / WN_HH, shl = srow % WN_HH;
    const bool fuse_in = in_affine != nullptr;   // workgroup-uniform
    const float relu_floor = in_relu ? 0.0f : -3.0e38f;   // (the fused ReLU as a max against a uniform: no branch in the staging code)
    int aff_n = -1;
    auto row_off = [&]() -> unsigned {   // byte offset of (n, channel 2cp of chunk 0, row, w0), or OOB (padding rows / idle threads)
        const int gd = d0 - 1 + sdl, gh = h0 - 1 + shl;
        if (stager && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H)
            return (unsigned)((((long)n * Ci + 2 * cp) * DHW + (long)gd * HW + gh * W + w0) * 4);
        return OOB;
    };
    f32x4 xa0, xb0, xa1, xb1;   // channels 2cp / 2cp+1: voxels w0..w0+3, w0+4..w0+7
    float xl0, xr0, xl1, xr1;   // ... w0-1, w0+8
#define WN_LOAD_X(chunk)                                                                                   \
    {                                                                                                      \
        const unsigned soff_ = (unsigned)((long)(chunk) * WN_KC * DHW * 4);                                \
        const unsigned o_ = row_off();                                                        \
        const unsigned o1_ = o_ == OOB ? OOB : o_ + chan_stride;                                           \
        const bool lft_ = o_ != OOB && w0 > 0, rgt_ = o_ != OOB && w0 + WN_TW < W;                         \
        xa0 = buf_load_f4(rsrc, o_, soff_);                                                                \
        xb0 = buf_load_f4(rsrc, o_ == OOB ? OOB : o_ + 16u, soff_);                                        \
        xa1 = buf_load_f4(rsrc, o1_, soff_);                                                               \
        xb1 = buf_load_f4(rsrc, o1_ == OOB ? OOB : o1_ + 16u, soff_);                                      \
        xl0 = buf_load_f(rsrc, lft_ ? o_ - 4u : OOB, soff_);                                               \
        xr0 = buf_load_f(rsrc, rgt_ ? o_ + 32u : OOB, soff_);                                              \
        xl1 = buf_load_f(rsrc, lft_ ? o1_ - 4u : OOB, soff_);                                              \
        xr1 = buf_load_f(rsrc, rgt_ ? o1_ + 32u : OOB, soff_);                                             \
    }
    unsigned xmax_ = 0;   // max |scaled halo value| this thread staged, as bits (NaN sorts above Inf above finite): beyond 2^15 the
                          // transformed values can leave the f16 range
    // registers -> (fused GroupNorm + ReLU) -> scale -> F(2,3) input transform -> split -> LDS
#define WN_STAGER_ON stager
    // All arithmetic on (channel 2cp, channel 2cp+1) PAIRS (f32x2): the pair is exactly the half2 a staging store writes, so scale,
    // transform and split run on v_pk_mul_f32 / v_pk_add_f32 / v_cvt_pk_f16_f32 — ~130 VALU instructions per thread and chunk instead of
    // ~560 with scalar conversions (the halo write was 21 % of a launch: profiles/r04_wino_ablations.txt).  Split: hi = rne(t), lo =
    // rne(t - hi); a non-finite t gives hi = Inf / NaN and lo = NaN, i.e. a non-finite product, like the reference's fp32 conv.
#define WN_WRITE_X(chunk, FUSE)                                                                            \
    if (WN_STAGER_ON) {                                                                                    \
        f32x2 v_[10] = {{xl0, xl1}, {xa0[0], xa1[0]}, {xa0[1], xa1[1]}, {xa0[2], xa1[2]}, {xa0[3], xa1[3]},  \
                        {xb0[0], xb1[0]}, {xb0[1], xb1[1]}, {xb0[2], xb1[2]}, {xb0[3], xb1[3]}, {xr0, xr1}}; \
        if (FUSE) {   /* compile-time.  Padding (rows / edge voxels outside the volume) must stay 0: its (scale, shift) pair is zeroed, */ \
                      /* and max(0*x + 0, floor) = 0 for both floors — no select, no branch                                       */ \
            const int gd_ = d0 - 1 + sdl, gh_ = h0 - 1 + shl;                                              \
            const float mid_ = ((unsigned)gd_ < (unsigned)D && (unsigned)gh_ < (unsigned)H) ? 1.0f : 0.0f; \
            const float lft_ = w0 > 0 ? mid_ : 0.0f, rgt_ = w0 + WN_TW < W ? mid_ : 0.0f;                  \
            const float4 sc_ = *reinterpret_cast<const float4 *>(aff + ((chunk) * WN_KC + 2 * cp) * 2);    \
            const f32x2 mul_ = {sc_.x, sc_.z}, add_ = {sc_.y, sc_.w};                                      \
            const f32x2 mm_ = mul_ * mid_, am_ = add_ * mid_, ml_ = mul_ * lft_, al_ = add_ * lft_, mr_ = mul_ * rgt_, ar_ = add_ * rgt_; \
            _Pragma("unroll") for (int i = 0; i < 10; ++i) {                                               \
                const f32x2 a_ = v_[i] * (i == 0 ? ml_ : i == 9 ? mr_ : mm_) + (i == 0 ? al_ : i == 9 ? ar_ : am_); \
                v_[i][0] = fmaxf(a_[0], relu_floor);                                                       \
                v_[i][1] = fmaxf(a_[1], relu_floor);                                                       \
            }                                                                                              \
        }                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 10; ++i) {                                                   \
            v_[i] *= x_scale;                                                                              \
            xmax_ = max(xmax_, max(__float_as_uint(v_[i][0]) & 0x7fffffffu, __float_as_uint(v_[i][1]) & 0x7fffffffu));   /* |bits|: NaN > Inf > finite */ \
        }                                                                                                  \
        _Float16 *xd_ = Xs + (cp >> 2) * WN_XBLK + srow * 32 + (cp & 3) * 2;                               \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                    \
            const f32x2 t_[4] = {v_[2 * q] - v_[2 * q + 2], v_[2 * q + 1] + v_[2 * q + 2], v_[2 * q + 2] - v_[2 * q + 1],        \
                                 v_[2 * q + 1] - v_[2 * q + 3]};                                           \
            _Pragma("unroll") for (int pp = 0; pp < 4; ++pp) {                                             \
                const half2v hv_ = __builtin_convertvector(t_[pp], half2v);                                \
                const half2v lv_ = __builtin_convertvector(t_[pp] - __builtin_convertvector(hv_, f32x2), half2v); \
                *reinterpret_cast<half2v *>(xd_ + pp * 2 * WN_XBLK + q * 8) = hv_;                         \
                *reinterpret_cast<half2v *>(xd_ + WN_XPART + pp * 2 * WN_XBLK + q * 8) = lv_;              \
            }                                                                                              \
            __builtin_amdgcn_sched_barrier(0);   /* one output pair at a time: hoisting all 32 conversions above the stores spills */ \
        }                                                                                                  \
    }

    // ---- weight stream: slab s of this workgroup = (chunk (s / 9) % nchunks, group s % 9), ring slot s & 3 ---------------------
    const unsigned ws_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) _Float16 *)Ws;
    const _Float16 *const wsrc = wslabs + (size_t)cot * nchunks * WN_NG * WN_SLAB_HALFS + (size_t)lane * 8;
    int dma_s = 0, dma_cg = c_begin * WN_NG;   // slabs issued so far; its (chunk*9 + group) index into the packed tensor
    const int cg_total = c_end * WN_NG;
    auto dma_issue = [&]() -> int {   // 3 pieces of 1 KiB per wave; returns the number of vector-memory instructions issued
        if (dma_s >= s_total) return 0;
        const _Float16 *src = wsrc + (size_t)dma_cg * WN_SLAB_HALFS + wave * 512;
        const unsigned dst = ws_lds + (unsigned)(dma_s & (WN_RING - 1)) * (WN_SLAB_HALFS * 2) + (unsigned)wave * 1024u;
#pragma unroll
        for (int q = 0; q < 3; ++q) lds_dma16(src + q * 8 * 512, dst + (unsigned)q * 8192u);
        ++dma_s;
        if (++dma_cg == cg_total) dma_cg = c_begin * WN_NG;
        return 3;
    };

    // fragment bases (halfs)
    const int a_base = ((p * 2 + kgl) * WN_COT + j) * 8;   // + part*WN_PART_HALFS + m*256, inside a ring slot
    int b_base[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) b_base[t] = (p * 2 + kgl) * WN_XBLK + (((2 * ch + t) * WN_HH + (j >> 2)) * 4 + (j & 3)) * 8;

    auto load_aff = [&]() {
        for (int i = tid; i < Ci * 2; i += 512) aff[i] = in_affine[(size_t)n * Ci * 2 + i];
        aff_n = n;
    };

    WPROF_DECL;
    WN_STAMP(0)
    [[maybe_unused]] int wn_tile_ = 1;
    // ---- prologue: slabs 0..2 in flight, X(chunk 0) staged ---------------------------------------------------------------------
    if (fuse_in) {
        load_aff();
        lds_barrier();
    }
    dma_issue();
    dma_issue();
    dma_issue();
    WN_LOAD_X(c_begin);
    if (fuse_in) { WN_WRITE_X(c_begin, true) } else { WN_WRITE_X(c_begin, false) }
    lds_dma_wait<0>();
    lds_barrier();

    half8 ah[3], al[3], bl[2], bh[2][2];
    int s = 0;   // slab being consumed
#pragma unroll
    for (int m = 0; m < 3; ++m) al[m] = *reinterpret_cast<const half8 *>(Ws + a_base + WN_PART_HALFS + m * 256);
#pragma unroll
    for (int t = 0; t < 2; ++t) bh[0][t] = *reinterpret_cast<const half8 *>(Xs + b_base[t]);

    const float unscale = whdr[0] * x_unscale;
    const int co0 = cot * WN_COT;
    const bool direct = gridDim.z == 1;
    WPROF_ADD(0)

    for (int tj = j_first; tj < ntiles; tj += (int)gridDim.x) {
        const int en = n, ed0 = d0, eh0 = h0, ew0 = w0, etile = tile_at(tj);   // this tile (the staging variables move on during its last chunk)
        const bool has_next = tj + (int)gridDim.x < ntiles;
        int tz = 0;
        asm volatile("" : "+v"(tz));  // opaque 0, new per tile: keeps the epilogue's per-channel address math / bias loads from being hoisted
                                      // out of the tile loop into registers (where they were spilled to scratch)
        f32x16 acc[3][2];
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.0f;

        for (int c = c_begin; c < c_end; ++c) {
            const bool more = c + 1 < c_end;
            const bool do_load = more || has_next;   // a halo prefetch is issued in this chunk's interval 2 (uniform)
            const int load_chunk = more ? c + 1 : c_begin;
            if (!more && has_next) {
                decode_tile(tile_at(tj + (int)gridDim.x));   // staging now addresses the next tile
                // (the affine table was last read by the WRITE_X that ended the previous chunk, a barrier ago)
                if (fuse_in && n != aff_n) load_aff();
            }
            // One interval = one (kd,kh) slab: 18 MFMAs per wave (P1 = Wlo*Xhi, P2 = Whi*Xhi, P3 = Whi*Xlo).
            //   top      : DMA of slab s+3 into the slot slab s-1 left (every wave is past barrier s-1); interval 2 first issues the
            //              halo prefetch of the next chunk
            //   fragments: Whi(s), Xlo now; Wlo(s+1) and Xhi(next tap) after P1 — slab s+1 was published by barrier s-1
            //   bottom   : wait for THIS wave's pieces of slab s+2 (issued one interval ago; younger transfers stay in flight),
            //              barrier s: slab s+2 published, slot of slab s free
#define WN_TOFF(G) ((((G) / 3) * WN_HH + (G) % 3) * 32)
#define WN_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x16_f16(a_, b_, c_, 0, 0, 0)
#ifdef MPHIP_WN_DMA_FIRST   /* dev: same-box A/B — r04's first order: DMA issue and fragment reads in front of the interval's first MFMAs */
#define WN_INTERVAL(G)                                                                                                     \
    {                                                                                                                      \
        constexpr int cur_ = (G) & 1;                                                                                      \
        if ((G) == 2 && do_load) { WN_LOAD_X(load_chunk); }   /* (before the DMA: hipcc's own vmcnt(k) waits on the   */ \
        const int issued_ = dma_issue();                       /*  registers it reloads then only cover landed pieces) */ \
        const _Float16 *wsb_ = Ws + (s & (WN_RING - 1)) * WN_SLAB_HALFS + a_base;                                          \
        const _Float16 *wsn_ = Ws + ((s + 1) & (WN_RING - 1)) * WN_SLAB_HALFS + a_base + WN_PART_HALFS;                    \
        _Pragma("unroll") for (int m = 0; m < 3; ++m) ah[m] = *reinterpret_cast<const half8 *>(wsb_ + m * 256);            \
        _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                                      \
            bl[t] = *reinterpret_cast<const half8 *>(Xs + WN_XPART + b_base[t] + WN_TOFF(G));                              \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        _Pragma("unroll") for (int m = 0; m < 3; ++m)                                                                      \
            _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                                  \
                acc[m][t] = WN_MFMA(al[m], bh[cur_][t], acc[m][t]);                \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        if (s + 1 < s_total) {                                                                                             \
            _Pragma("unroll") for (int m = 0; m < 3; ++m) al[m] = *reinterpret_cast<const half8 *>(wsn_ + m * 256);        \
        }                                                                                                                  \
        if ((G) < 8) {                                                                                                     \
            _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                                  \
                bh[cur_ ^ 1][t] = *reinterpret_cast<const half8 *>(Xs + b_base[t] + WN_TOFF(((G) < 8 ? (G) + 1 : 0)));     \
        }                                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        _Pragma("unroll") for (int m = 0; m < 3; ++m)                                                                      \
            _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                                  \
                acc[m][t] = WN_MFMA(ah[m], bh[cur_][t], acc[m][t]);                \
        _Pragma("unroll") for (int m = 0; m < 3; ++m)                                                                      \
            _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                                  \
                acc[m][t] = WN_MFMA(ah[m], bl[t], acc[m][t]);                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        WPROF_ADD(1)                                                                                                       \
        /* younger than this wave's pieces of slab s+2: interval 2's halo prefetch (8) and this interval's pieces (3) */   \
        if ((G) == 2 && do_load) {                                                                                         \
            if (issued_) lds_dma_wait<WN_XLOADS + 3>(); else lds_dma_wait<WN_XLOADS>();                                    \
        } else {                                                                                                           \
            if (issued_) lds_dma_wait<3>(); else lds_dma_wait<0>();                                                        \
        }                                                                                                                  \
        WPROF_ADD(2)                                                                                                       \
        lds_barrier();                                                                                                     \
        WPROF_ADD(3)                                                                                                       \
        ++s;                                                                                                               \
    }
#else
            // The interval's first six MFMAs (Wlo(s) x Xhi: both fetched during the previous interval) are issued straight after the
            // barrier; the DMA pieces, the halo prefetch and all of the interval's fragment reads are issued into their shadow — an LDS-DMA
            // piece costs its wave 60-185 issue cycles (MI355X_MICROARCH.md), which the MFMA pipe sat out when they came first.
#define WN_INTERVAL(G)                                                                                                     \
    {                                                                                                                      \
        constexpr int cur_ = (G) & 1;                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        _Pragma("unroll") for (int m = 0; m < 3; ++m)                                                                      \
            _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                                  \
                acc[m][t] = WN_MFMA(al[m], bh[cur_][t], acc[m][t]);                \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        const _Float16 *wsb_ = Ws + (s & (WN_RING - 1)) * WN_SLAB_HALFS + a_base;                                          \
        const _Float16 *wsn_ = Ws + ((s + 1) & (WN_RING - 1)) * WN_SLAB_HALFS + a_base + WN_PART_HALFS;                    \
        _Pragma("unroll") for (int m = 0; m < 3; ++m) ah[m] = *reinterpret_cast<const half8 *>(wsb_ + m * 256);            \
        _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                                      \
            bl[t] = *reinterpret_cast<const half8 *>(Xs + WN_XPART + b_base[t] + WN_TOFF(G));                              \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        if ((G) == 2 && do_load) { WN_LOAD_X(load_chunk); }   /* (before the DMA: hipcc's own vmcnt(k) waits on the   */ \
        const int issued_ = dma_issue();                       /*  registers it reloads then only cover landed pieces) */ \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        if (s + 1 < s_total) {                                                                                             \
            _Pragma("unroll") for (int m = 0; m < 3; ++m) al[m] = *reinterpret_cast<const half8 *>(wsn_ + m * 256);        \
        }                                                                                                                  \
        if ((G) < 8) {                                                                                                     \
            _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                                  \
                bh[cur_ ^ 1][t] = *reinterpret_cast<const half8 *>(Xs + b_base[t] + WN_TOFF(((G) < 8 ? (G) + 1 : 0)));     \
        }                                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        _Pragma("unroll") for (int m = 0; m < 3; ++m)                                                                      \
            _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                                  \
                acc[m][t] = WN_MFMA(ah[m], bh[cur_][t], acc[m][t]);                \
        _Pragma("unroll") for (int m = 0; m < 3; ++m)                                                                      \
            _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                                  \
                acc[m][t] = WN_MFMA(ah[m], bl[t], acc[m][t]);                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        WPROF_ADD(1)                                                                                                       \
        /* younger than this wave's pieces of slab s+2: interval 2's halo prefetch (8) and this interval's pieces (3) */   \
        if ((G) == 2 && do_load) {                                                                                         \
            if (issued_) lds_dma_wait<WN_XLOADS + 3>(); else lds_dma_wait<WN_XLOADS>();                                    \
        } else {                                                                                                           \
            if (issued_) lds_dma_wait<3>(); else lds_dma_wait<0>();                                                        \
        }                                                                                                                  \
        WPROF_ADD(2)                                                                                                       \
        lds_barrier();                                                                                                     \
        WPROF_ADD(3)                                                                                                       \
        ++s;                                                                                                               \
    }
#endif
            WN_INTERVAL(0)
            WN_INTERVAL(1)
            WN_INTERVAL(2)
            WN_INTERVAL(3)
            WN_INTERVAL(4)
            WN_INTERVAL(5)
            WN_INTERVAL(6)
            WN_INTERVAL(7)
#ifdef MPHIP_WN_NO_OVERLAP8   /* dev: same-box A/B — the halo write in its own phase after the chunk's last interval */
            WN_INTERVAL(8)
            if (more) {
                if (fuse_in) { WN_WRITE_X(c + 1, true) } else { WN_WRITE_X(c + 1, false) }   // every wave is past its last read of the X tile
                WPROF_ADD(4)
                lds_barrier();
#pragma unroll
                for (int t = 0; t < 2; ++t) bh[0][t] = *reinterpret_cast<const half8 *>(Xs + b_base[t]);
                WPROF_ADD(5)
            }
#else
            {
                // The chunk's last interval carries the NEXT chunk's halo write: once this interval's fragments are in registers nobody reads
                // the X tile any more, so after one early barrier the tile may be rewritten while the interval's 18 MFMAs run.  The two
                // waves of a SIMD — (p, ch = 0) and (p, ch = 1): a workgroup's waves go round the SIMDs, wave w and w + 4 meet — take the
                // two jobs in OPPOSITE order: one issues its MFMAs while the other transforms, splits and stores its share of the halo
                // (VALU + LDS), then they swap.  The MFMAs stay ONE straight-line copy (accumulators that merge from two branches cost
                // hipcc 70-160 spilled registers, and so did interleaving both jobs in one instruction stream); only the staging code
                // sits under the wave-uniform conditions.  A tile's last chunk (!more) has nothing to stage here: the next tile's halo is
                // written after the output transform, which uses the X region.
#ifndef MPHIP_WN_DMA_FIRST
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[m][t] = WN_MFMA(al[m], bh[0][t], acc[m][t]);
                __builtin_amdgcn_sched_barrier(0);
#endif
                const int issued_ = dma_issue();
                const _Float16 *wsb_ = Ws + (s & (WN_RING - 1)) * WN_SLAB_HALFS + a_base;
                const _Float16 *wsn_ = Ws + ((s + 1) & (WN_RING - 1)) * WN_SLAB_HALFS + a_base + WN_PART_HALFS;
#pragma unroll
                for (int m = 0; m < 3; ++m) ah[m] = *reinterpret_cast<const half8 *>(wsb_ + m * 256);
#pragma unroll
                for (int t = 0; t < 2; ++t) bl[t] = *reinterpret_cast<const half8 *>(Xs + WN_XPART + b_base[t] + WN_TOFF(8));
                if (more) {
                    lds_barrier();   // every wave holds its last X fragments: the tile may be rewritten
                    WPROF_ADD(3)
                    if (ch != 0) {
                        if (fuse_in) { WN_WRITE_X(c + 1, true) } else { WN_WRITE_X(c + 1, false) }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#ifdef MPHIP_WN_DMA_FIRST
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[m][t] = WN_MFMA(al[m], bh[0][t], acc[m][t]);
                __builtin_amdgcn_sched_barrier(0);
#endif
                if (s + 1 < s_total) {
#pragma unroll
                    for (int m = 0; m < 3; ++m) al[m] = *reinterpret_cast<const half8 *>(wsn_ + m * 256);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[m][t] = WN_MFMA(ah[m], bh[0][t], acc[m][t]);
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[m][t] = WN_MFMA(ah[m], bl[t], acc[m][t]);
                __builtin_amdgcn_sched_barrier(0);
                if (more && ch == 0) {
                    if (fuse_in) { WN_WRITE_X(c + 1, true) } else { WN_WRITE_X(c + 1, false) }
                }
                __builtin_amdgcn_sched_barrier(0);
                WPROF_ADD(1)
                if (issued_) lds_dma_wait<3>(); else lds_dma_wait<0>();
                WPROF_ADD(2)
                lds_barrier();   // slab s+2 published, slot of slab s free, the new halo visible
                ++s;
                if (more) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) bh[0][t] = *reinterpret_cast<const half8 *>(Xs + b_base[t]);
                }
                WPROF_ADD(5)
            }
#endif
#undef WN_INTERVAL
        }

        // ---- output transform + epilogue: three rounds (one 32-channel row tile each) through the dead X region ---------------------
        const int gn_rows = tiles_total * 2;   // channel-major [Co][tile * 2 + ch][2] (the finalize kernel reads rows of it)
        if (gn_part && etile == 0 && tid == 0) gn_part[(size_t)gn_rows * Co * 2] = unscale;   // (behind the partials)
        const bool odd = (lane & 1) != 0;
        // (row start of this lane's QUAD of voxels: lanes 2k / 2k+1 store the 4 voxels 4k..4k+3 of a row, for different channels)
        float *const dsto = (direct ? y : y + (size_t)blockIdx.z * N * Co * DHW) + (size_t)en * Co * DHW + (size_t)(ed0 + 2 * ch) * HW + (size_t)(eh0 + (j >> 2)) * W + ew0 + 2 * (j & 2);
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            // park the units other waves finish: unit u = accumulator registers 4u..4u+3 of both column tiles; wave p keeps unit p
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (u != p) {
                        const int slot = t * 3 + (u - (u > p ? 1 : 0));
                        const f32x4 v = {acc[m][t][4 * u], acc[m][t][4 * u + 1], acc[m][t][4 * u + 2], acc[m][t][4 * u + 3]};
                        *reinterpret_cast<f32x4 *>(Ex + ((wave * 6 + slot) * 64 + lane) * 4) = v;
                    }
            lds_barrier();
            float ssum[4] = {0.0f, 0.0f, 0.0f, 0.0f}, qsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            float bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) bv[i] = (direct && bias) ? bias[co0 + m * 32 + 8 * p + 4 * kgl + i + tz] : 0.0f;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 M[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q != p) {
                        const int slot = t * 3 + (p - (p > q ? 1 : 0));
                        M[q] = *reinterpret_cast<const f32x4 *>(Ex + (((ch * 4 + q) * 6 + slot) * 64 + lane) * 4);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) M[q][i] = acc[m][t][4 * q + i];
                    }
                }
                float y0[4], y1[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float r0 = (M[0][i] + M[1][i]) + M[2][i];
                    const float r1 = (M[1][i] - M[2][i]) - M[3][i];
                    ssum[i] += r0 + r1;
                    qsum[i] = __builtin_fmaf(r0, r0, qsum[i]);
                    qsum[i] = __builtin_fmaf(r1, r1, qsum[i]);
                    y0[i] = r0 * unscale + bv[i];
                    y1[i] = r1 * unscale + bv[i];
                }
                // 16-byte stores: a lane holds one output pair (2 voxels) of 4 channels; lanes 2k / 2k+1 hold neighbouring pairs of a row.
                // They trade halves (quad_perm [1,0,3,2]): the even lane ends up with 4 consecutive voxels of channels 0-1, the odd lane
                // with those of channels 2-3 — two dwordx4 stores per lane instead of four dwordx2 (the epilogue is store-ISSUE bound:
                // 8-byte stores of 32-byte row pieces ran at ~7 B/clk/CU, MI355X_MICROARCH.md "epilogue store tail").
#define WN_SWAP(v_) __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v_), 0xB1, 0xf, 0xf, false))
                const float g0 = WN_SWAP(odd ? y0[0] : y0[2]), g1 = WN_SWAP(odd ? y1[0] : y1[2]);
                const float g2 = WN_SWAP(odd ? y0[1] : y0[3]), g3 = WN_SWAP(odd ? y1[1] : y1[3]);
#undef WN_SWAP
                const f32x4 va = {odd ? g0 : y0[0], odd ? g1 : y1[0], odd ? y0[2] : g0, odd ? y1[2] : g1};
                const f32x4 vb = {odd ? g2 : y0[1], odd ? g3 : y1[1], odd ? y0[3] : g2, odd ? y1[3] : g3};
                float *const dq = dsto + (size_t)(co0 + m * 32 + 8 * p + 4 * kgl + (odd ? 2 : 0) + tz) * DHW + (size_t)t * HW;
                *reinterpret_cast<f32x4 *>(dq) = va;
                *reinterpret_cast<f32x4 *>(dq + DHW) = vb;
            }
            if (gn_part) {
                // per-channel (sum, sum of squares) of the RAW transformed accumulators over this wave's 2 x 64 voxels of the channel:
                // the 32 lanes of a half-wave hold one channel's columns (the finalize kernel applies unscale and the bias in double)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#define WN_ROW_ADD(v_, ctrl_) v_ += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v_), ctrl_, 0xf, 0xf, false));
                    WN_ROW_ADD(ssum[i], 0x128) WN_ROW_ADD(qsum[i], 0x128)   // row_ror:8, :4, :2, :1 -> every lane of a 16-lane row: the row's sum
                    WN_ROW_ADD(ssum[i], 0x124) WN_ROW_ADD(qsum[i], 0x124)
                    WN_ROW_ADD(ssum[i], 0x122) WN_ROW_ADD(qsum[i], 0x122)
                    WN_ROW_ADD(ssum[i], 0x121) WN_ROW_ADD(qsum[i], 0x121)
#undef WN_ROW_ADD
                    ssum[i] += __shfl_xor(ssum[i], 16, 64);
                    qsum[i] += __shfl_xor(qsum[i], 16, 64);
                }
                if (j == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int co = co0 + m * 32 + 8 * p + 4 * kgl + i + tz;
                        *reinterpret_cast<float2 *>(gn_part + ((size_t)co * gn_rows + (size_t)etile * 2 + ch) * 2) = make_float2(ssum[i], qsum[i]);
                    }
                }
            }
            lds_barrier();   // the region is rewritten by the next round / the next tile's halo
        }

        WPROF_ADD(6)
        WN_STAMP(wn_tile_) ++wn_tile_;
        if (has_next) {
            if (fuse_in) { WN_WRITE_X(c_begin, true) } else { WN_WRITE_X(c_begin, false) }   // the next tile's first halo chunk (prefetched during this tile's last chunk)
            WPROF_ADD(4)
            lds_barrier();
#pragma unroll
            for (int t = 0; t < 2; ++t) bh[0][t] = *reinterpret_cast<const half8 *>(Xs + b_base[t]);
            WPROF_ADD(5)
        }
    }
#undef WN_LOAD_X
#undef WN_WRITE_X
#undef WN_TOFF
    // operands outside the f16 range (non-finite inputs, or finite ones beyond a wrong caller-supplied descriptor) are not clamped — they
    // propagate as Inf / NaN — but they are counted: here per thread that saw any (the direct kernel counts elements)
    const unsigned xlim_ = __float_as_uint(0.5f * F16_CLAMP);
    if (__builtin_amdgcn_ballot_w64(xmax_ > xlim_) != 0) {  // never taken in normal operation
        unsigned tot = xmax_ > xlim_;
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) tot += __shfl_xor(tot, sft, 64);
        if (lane == 0) atomicAdd(&g_f16x3_wino_saturated, (unsigned long long)tot);
    }
    WPROF_FLUSH
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
int f16x3_wino_saturation(unsigned long long *count, int reset) {   // (mphip_f16x3_saturation_count adds it to the direct kernels' counter)
    if (hipMemcpyFromSymbol(count, HIP_SYMBOL(g_f16x3_wino_saturated), sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long pp = 0;   // (the role-split kernel keeps its own counter: separate translation unit)
    if (f16x3_wino_pp_saturation(&pp, reset) != 0) return -1;
    *count += pp;
    if (f16x3_wino_bt_saturation(&pp, reset) != 0) return -1;
    *count += pp;
    if (reset) {
        const unsigned long long z = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_f16x3_wino_saturated), &z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}

static bool wino_enabled() {   // dev: same-box A/B against the direct kernel (read per call: tests flip it in-process; packs always carry the slabs)
    const char *e = getenv("MPHIP_WINOGRAD");
    return !(e && e[0] == '0');
}

// layers that can ever take the transformed-domain kernel get its slabs behind the direct pack (a weight tensor does not know the
// volume it will meet): Ci <= 384 covers G3d's levels 0-2 and Eapp's 3-D tail, +133 % pack bytes on <= 16 MB tensors
size_t f16x3_wino_packed_bytes(int Co, int Ci) {
    static const bool nopack = getenv("MPHIP_WINOGRAD_PACK") && getenv("MPHIP_WINOGRAD_PACK")[0] == '0';   // dev: bisecting (process-wide: set before the first pack)
    if (nopack) return 0;
    // (Ci <= 768: G3d's 2x8x8 level — 384 / 768 channels — takes the two-frame mode of the big-tile kernel, r06; the LDS table of the fused
    //  input GroupNorm limits the 4-plane kernels to Ci <= 384 only when the norm is fused, checked at launch)
    if (Ci % WN_KC || Co % WN_COT || Ci > 2 * WN_AFF_CI) return 0;
    return (size_t)(Co / WN_COT) * (Ci / WN_KC) * WN_NG * WN_SLAB_HALFS * sizeof(_Float16);
}

// split-K factor of a launch (whole chunks only): the largest divisor of the chunk count that keeps the launch inside ONE round of
// resident workgroups (one per CU) — the direct kernel's r03 rule
// tiles of a launch: 4 x 8 x 8 voxels of one frame, or — depth-2 volumes, conv3d_f16x3_wino_bt.hip's D2 mode — 2 x 8 x 8 voxels of TWO frames
long f16x3_wino_tiles(int N, int D, int H, int W) {
    return D == 2 ? (long)((N + 1) / 2) * (H / WN_TH) * (W / WN_TW) : (long)N * (D / WN_TD) * (H / WN_TH) * (W / WN_TW);
}

int f16x3_wino_splits(int N, int Ci, int Co, int D, int H, int W) {
    const long base = f16x3_wino_tiles(N, D, H, W) * (Co / WN_COT);
    const int nchunks = Ci / WN_KC;
    int sp = 1;
    if (base < 256)
        for (int dv = 2; dv <= nchunks; ++dv)
            if (nchunks % dv == 0 && base * dv <= 256) sp = dv;
    return sp;
}

bool f16x3_wino_usable(int N, int Ci, int Co, int D, int H, int W) {
    const char *d2_env = getenv("MPHIP_WINOGRAD_D2");   // dev: same-box A/B of the two-frame mode against the direct kernel (read per call: tests flip it in-process)
    const bool d2_off = d2_env && d2_env[0] == '0';
    if (!wino_enabled() || f16x3_wino_packed_bytes(Co, Ci) == 0 || (D % WN_TD && (D != 2 || d2_off)) || H % WN_TH || W % WN_TW) return false;
    // (two frames per tile: a single frame leaves half of every tile empty — B = 1, 768 -> 768: 31.6 us against the direct kernel's 28.8;
    //  from B = 4, the training shard, the mode wins: tools/d2_check.py)
    if (D == 2 && N < 4 && !getenv("MPHIP_WINOGRAD_MIN_TILES")) return false;
    if (D != 2 && Ci > WN_AFF_CI) return false;   // (the 4-plane kernels keep r05's range: their launches may fuse the input GroupNorm through the LDS table)
    // one workgroup per CU, ~1 us per (kd,kh) slab: worth it when the launch (with its split-K factor) fills the chip and the direct
    // kernel's advantage — a 512-voxel tile's weight economy, thirds of a tile per CU — does not apply (measured: tools/wino_check.py)
    const long tiles = f16x3_wino_tiles(N, D, H, W);
    const char *min_s = getenv("MPHIP_WINOGRAD_MIN_TILES");   // dev: threshold sweep
    const long min_wgs = min_s ? atol(min_s) : 192;
    return tiles * (Co / WN_COT) * f16x3_wino_splits(N, Ci, Co, D, H, W) >= min_wgs;
}

int f16x3_wino_launch(const float *x, const void *slabs, const float *hdr, const float *bias, float *dst, int N, int Ci, int Co, int D,
                      int H, int W, int splits, const float *in_affine, int in_relu, const float *x_range, hipStream_t s, const int *tile_list,
                      float *gn_part, hipEvent_t t0, hipEvent_t t1) {
    const int tiles = (int)f16x3_wino_tiles(N, D, H, W), cots = Co / WN_COT;
    static const char *cus_s = getenv("MPHIP_CONV_CUS");   // dev: persistent grid size (leave CUs to another batch's small kernels)
    const long cus = cus_s ? atol(cus_s) : 256;
    long gx = (cus + cots * splits - 1) / (cots * splits);   // persistent: one workgroup per CU
    if (gx > tiles) gx = tiles;
    const dim3 grid((unsigned)gx, (unsigned)cots, (unsigned)splits);
    const int cps = (Ci / WN_KC + splits - 1) / splits;
    const unsigned xb = (unsigned)((size_t)N * Ci * D * H * W * 4);
    static const int xcd_on = !(getenv("MPHIP_F16X3_XCD") && getenv("MPHIP_F16X3_XCD")[0] == '0');
    const char *pp_env = getenv("MPHIP_WINO_PP");   // dev: same-box A/B against the lockstep schedule (read per call: tools flip it in-process)
    const bool pp_on = !(pp_env && pp_env[0] == '0');
    // MPHIP_WINO_PP: 0 the lockstep kernel (r04), 1 the role-split kernel (r05), 2 the big-tile kernel (r06: one wave per SIMD; bit-identical
    // to 1).  The one-product (autocast) arithmetic exists on the role-split schedule only.
    if (D == 2) {   // the two-frame mode exists in the big-tile kernel only (three-product arithmetic; under the autocast policy as well:
                    // this level is 4 % of the slice's multiplies)
        if (tile_list || gn_part) {
            set_error("conv3d_fwd(f16x3, F(2,3), D = 2): no demand-driven tile list / GroupNorm partials in the two-frame mode");
            return MPHIP_EINVAL;
        }
        f16x3_wino_bt_launch(grid, s, t0, t1, x, (const _Float16 *)slabs, hdr, bias, dst, N, Ci, Co, D, H, W, cps, xb, in_affine, in_relu, x_range,
                             tiles, xcd_on, tile_list, gn_part);
        return check_launch("conv3d_fwd(f16x3, F(2,3), big tile, two frames)");
    }
    const bool bt_on = pp_env && pp_env[0] == '2' && !conv_half_products();
    if (bt_on) {
        f16x3_wino_bt_launch(grid, s, t0, t1, x, (const _Float16 *)slabs, hdr, bias, dst, N, Ci, Co, D, H, W, cps, xb, in_affine, in_relu, x_range,
                             tiles, xcd_on, tile_list, gn_part);
        return check_launch("conv3d_fwd(f16x3, F(2,3), big tile)");
    }
    if (pp_on) {   // the role-split schedule (conv3d_f16x3_wino_pp.hip): same arithmetic, same packed weights, same tile
        f16x3_wino_pp_launch(grid, s, t0, t1, x, (const _Float16 *)slabs, hdr, bias, dst, N, Ci, Co, D, H, W, cps, xb, in_affine, in_relu, x_range,
                             tiles, xcd_on, tile_list, gn_part, conv_half_products());
        return check_launch("conv3d_fwd(f16x3, F(2,3), role-split)");
    }
    if (t0 && t1)
        hipExtLaunchKernelGGL(conv3d_k3_f16x3_wino_kernel, grid, dim3(512), 0, s, t0, t1, 0, x, (const _Float16 *)slabs, hdr, bias, dst, N, Ci,
                              Co, D, H, W, cps, xb, in_affine, in_relu, x_range, tiles, xcd_on, tile_list, gn_part);
    else
        hipLaunchKernelGGL(conv3d_k3_f16x3_wino_kernel, grid, dim3(512), 0, s, x, (const _Float16 *)slabs, hdr, bias, dst, N, Ci, Co, D, H, W,
                           cps, xb, in_affine, in_relu, x_range, tiles, xcd_on, tile_list, gn_part);
    return check_launch("conv3d_fwd(f16x3, F(2,3))");
}

}  // namespace mphip

#ifdef MPHIP_PROFILE_PHASES
extern "C" int mphip_debug_f16x3_wino_profile(unsigned long long *out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(mphip::g_f16x3_wino_prof), 64) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(mphip::g_f16x3_wino_prof), z, 64) != hipSuccess) return -1;
    }
    return 0;
}
#endif

#ifdef MPHIP_WN_TRACE
extern "C" int mphip_debug_wino_trace(unsigned long long *host_out /* 1024 * 16 */) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mphip::g_wn_trace), sizeof(unsigned long long) * 1024 * 16) == hipSuccess ? 0 : -1;
}
#endif
