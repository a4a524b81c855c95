// K4 fast mode, transformed domain, ONE wave per SIMD with a 96 x 128 register tile ("big tile") — the F(2,3) conv of
// conv3d_f16x3_wino.hip / conv3d_f16x3_wino_pp.hip: same arithmetic, same packed weights, same 4x8x8-voxel tile, same LDS map, and the
// SAME accumulation order per output element — its results are bit-identical to the role-split kernel's (tests hold them torch.equal).
//
// Why (r06): the role-split kernel's LOAD side, not the matrix pipe, bounds it.  Each of its eight waves multiplies a
// 96 x 64 tile (M = 96 output channels, N = 2 planes x 32 output pairs) and re-reads (96 + 64) x 16 x 2 B x (hi, lo) = 10 KB of fragments per
// K = 16 step — 80 KB per workgroup and step, 48 KB of it the SAME weight fragments read by the two waves of a SIMD (same Winograd
// position, different plane pairs).  Here a workgroup is 4 waves (one per SIMD, 512 registers each), wave = Winograd position, and a wave
// owns all four planes of the tile: 96 x 128 accumulators (192 AGPRs), 14 KB of fragments per step and wave, 56 KB per workgroup (-30 %),
// one barrier per step instead of four; with one wave per SIMD the overlap of loads and MFMAs is software pipelining inside the wave.
// MEASURED (profiles/NOTES_r06.md 3): bit-identical and SLOWER than the role-split kernel on the 4-plane launches (0.504 vs 0.444-0.452 ms):
// a single wave cannot hide its own issue stalls.  It is the default only of the TWO-FRAME MODE (D2 below: G3d's 2x8x8 level);
// MPHIP_WINO_PP=2 selects it elsewhere.
//
// Step s (K = 16: two (8-channel chunk, (kd,kh) tap) items, exactly the role-split kernel's K loop), per wave:
//      P1(s) = Wlo x Xhi | counted wait, BARRIER(s) | P2(s) = Whi x Xhi | P3(s) = Whi x Xlo          (12 MFMAs each)
// ONE fragment register set; every fragment is re-loaded in the segment after its last use, a full segment (>= 384 cycles) ahead of its
// next one:   during P1(s): Whi(s), Xlo(s)      during P2(s): Wlo(s+1)      during P3(s): Xhi(s+1).
// So slab s is read from P2(s-1) to P1(s) and is dead at BARRIER(s); slab s+1 must be visible there.  After the barrier a wave issues
// its six LDS-DMA pieces of slab s+3 into the slot of slab s (ring of 3), ONE per slot and >= 2 MFMAs apart, and waits for them just
// before BARRIER(s+2).
//
// X staging (the role-split kernel's 8-channel double buffer and pair-major image; both halves done by every thread as two "roles" with
// their own register sets):
//   role 0 -> buffer 0 (even chunk of the NEXT period; read until P1(4), next read P3(8)): loads in step 0, table prep in step 2,
//            conversion in step 3, the four output pairs written in steps 4-7;
//   role 1 -> buffer 1 (odd chunk; read from P3(3) to P1(8)): loads in step 3, prep in step 5, conversion in steps 6-7, pair 0 written in
//            step 8 and pairs 1-3 in steps 0-2 of the period the data belongs to.
// Every slice is cut into micro-operations of <= 7 instructions hung behind one MFMA each of P2 / P3 (pinned with sched_barrier: hipcc
// clusters otherwise), nothing is conditional (a branch costs the single wave two issue slots), P1 carries the fragment reads only, so
// that no LDS operation is young at the barrier.  The tile's epilogue goes through ONE exchange region (buffer 1 already holds the next
// period's first pair): six rounds (row tile x plane pair).
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include <hip/hip_ext.h>

#include "mphip_ablate.h"
#include "mphip_conv.h"
#include "mphip_f16x3.h"
#include "mphip_wino_tile.h"

namespace mphip {

constexpr int BT_PIECES = PP_SLAB_B / (4 * 1024);   // LDS-DMA pieces (1 KiB) of a slab per wave: 6

__device__ unsigned long long g_f16x3_wino_bt_saturated;

#ifdef MPHIP_BT_PROFILE
// dev instrumentation: shader cycles per wave and launch; [0] whole kernel, [1] epilogues, [2] waves, [3] prologue
__device__ unsigned long long g_bt_prof[8];
__device__ __forceinline__ unsigned long long bt_memtime() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#endif

// One LDS-DMA piece of 16 bytes per lane: global `base` (wave-uniform) + `off` (per lane) -> LDS byte address `lds` + 16 * lane (pp_dma3's
// contract: M0 written in the statement that reads it; `s_nop 4` for a base that may just have been reloaded from a spill lane).  One piece
// per statement: the step hangs each behind its own MFMA (a VMEM instruction holds the wave's issue for tens of cycles).
__device__ __forceinline__ void bt_dma1(const void *base, unsigned off, unsigned lds) {
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(lds) : "memory");
}

template <int N, class F, int... I>
__device__ __forceinline__ void bt_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void bt_for(F &&f) {
    bt_for_impl<N>(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}

// FUSE: the input GroupNorm (+ ReLU) is applied while staging (in_affine != nullptr) — a compile-time constant: a run-time flag costs the
// single wave a branch (two issue slots) in every conversion piece.
// D2: volumes of depth 2 (G3d's 2x8x8 level, model.py:576-589) — a tile is TWO frames x 2 planes x 8 x 8: the wave's four planes are
// (frame n: d 0, 1; frame n+1: d 0, 1), the six halo-plane slots hold [zero, n:d0, n:d1, zero, n+1:d0, n+1:d1] — plane t reads slots
// tb[t] + kd with tb = {0, 1, 3, 4}; the one combination that would leave the buffer (t = 3, kd = 2: the zero plane above frame n+1) reads
// slot 3, the other zero plane.  The fused-GroupNorm table of two frames does not fit the LDS: a thread loads its row's four values from
// global memory with its halo (one more load per role and period).  No demand-driven tile list, no GroupNorm partials in this mode.
template <bool FUSE, bool D2>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
conv3d_k3_f16x3_wino_bt_kernel(const float *__restrict__ x, const _Float16 *__restrict__ wslabs, const float *__restrict__ whdr,
                               const float *__restrict__ bias, float *__restrict__ y, int N, int Ci, int Co, int D, int H, int W,
                               int chunks_per_split, unsigned x_bytes, const float *__restrict__ in_affine, int in_relu,
                               const float *__restrict__ x_range, int tiles_total, int xcd_aware, const int *__restrict__ tile_list,
                               float *__restrict__ gn_part) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[PP_LDS_BYTES];
    float *const aff = reinterpret_cast<float *>(smem + PP_LDS_AFF);   // [Ci][2]: (scale, shift) of the fused input GroupNorm

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int p = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave = Winograd position
    const int j = lane & 31, kgl = lane >> 5;
    const int HW = H * W, DHW = D * HW;
    const int ntiles = tile_list ? tile_list[0] : tiles_total;
    auto tile_at = [&](int jj) -> int { return tile_list ? tile_list[1 + jj] : jj; };
    const int j_first = (tile_list || !xcd_aware) ? (int)blockIdx.x : (int)xcd_remap(blockIdx.x, gridDim.x);
    if (j_first >= ntiles) return;   // (workgroup-uniform, before any barrier)
    float x_scale = 16.0f, x_unscale = 1.0f / 16.0f;
    if (x_range) range_scale_block(x_range, x_scale, x_unscale);
#ifdef MPHIP_BT_PROFILE
    const unsigned long long prof_t0 = bt_memtime();
    unsigned long long prof_epi = 0;
#endif

    const int tiles_w = W / PP_TW, tiles_h = H / PP_TH, tiles_d = D / PP_TD;
    const int cot = blockIdx.y;
    const int nchunks = Ci / 16;
    const int c_begin = blockIdx.z * chunks_per_split, c_end = min(nchunks, c_begin + chunks_per_split);
    if (c_begin >= c_end) return;
    const int nper = c_end - c_begin;
    const int nmine = (ntiles - j_first + (int)gridDim.x - 1) / (int)gridDim.x;   // tiles this workgroup walks
    const int per_total = nmine * nper;                                           // 16-channel periods it walks

    auto period_at = [&](int tj, int chunk) -> PpPeriod {
        PpPeriod r;
        int bid = tile_at(tj);
        const int tw = bid % tiles_w; bid /= tiles_w;
        const int th = bid % tiles_h; bid /= tiles_h;
        if constexpr (D2) { r.n = 2 * bid; r.d0 = 0; }      // a tile = the frame pair (2 bid, 2 bid + 1)
        else { r.n = bid / tiles_d; r.d0 = (bid % tiles_d) * PP_TD; }
        r.h0 = th * PP_TH; r.w0 = tw * PP_TW;
        r.chunk = chunk; r.tj = tj;
        return r;
    };
    auto period_next = [&](const PpPeriod &a) -> PpPeriod {   // (only called when a successor exists)
        if (a.chunk + 1 < c_end) { PpPeriod r = a; r.chunk = a.chunk + 1; return r; }
        return period_at(a.tj + (int)gridDim.x, c_begin);
    };

    // raw buffer descriptor of x (base, stride 0, num_records = bytes, 32-bit float data format): out-of-range offsets read 0 = the padding
    const unsigned long long xaddr = (unsigned long long)(uintptr_t)x;
    const pp_u32x4 rsrc = {(unsigned)xaddr, (unsigned)(xaddr >> 32) & 0xffffu, x_bytes, 0x00020000u};
    const unsigned chan_stride = (unsigned)DHW * 4u;
    const unsigned long long aaddr = (unsigned long long)(uintptr_t)in_affine;
    const pp_u32x4 rsrc_aff = {(unsigned)aaddr, (unsigned)(aaddr >> 32) & 0xffffu, (unsigned)((size_t)N * Ci * 2 * 4), 0x00020000u};

    // ---- X staging: thread = (channel pair cp of an 8-channel half, halo row); 240 of the 256 threads (the other 16 re-do row 59:
    // same loads, same values, same LDS addresses — branch-free).  Role 0 = the even half (buffer 0), role 1 = the odd half (buffer 1):
    // two register sets, their loads ordered by hand (the counted wait before a step's barrier retires the loads of two steps ago).
    const int cp = tid & 3, srow = min(tid >> 2, PP_ROWS - 1);
    const int sdl = srow / PP_HH, shl = srow % PP_HH;
    constexpr bool fuse_in = FUSE;
    const float relu_floor = in_relu ? 0.0f : -3.0e38f;
    int aff_n = -1;
    f32x4 xa0[2], xb0[2], xa1[2], xb1[2];   // [role]: channels 2cp / 2cp+1 of the half: voxels w0..w0+3, w0+4..w0+7
    float xl0[2], xr0[2], xl1[2], xr1[2];   // ... w0-1, w0+8
    float xmaxf_ = 0.0f;                    // max |scaled halo value| this thread staged (finite or Inf) ...
    bool xnan_ = false;                     // ... and whether it saw a NaN (v_max drops them)
    unsigned t_row[2] = {OOB, OOB};         // (D2, fused) byte offset of the row's four table values ((scale, shift) of channels 2cp, 2cp+1)
    unsigned o_row[2] = {OOB, OOB};         // byte offset of (n, channel 2cp of the half, row, w0) of the unit being loaded, or OOB (padding rows)
    // ---- staging micro-operations: each is one piece of <= ~7 instructions that the step hangs behind one MFMA -----------------------------
    // (fused GroupNorm) per-role table values: (scale, shift) x operand scale S of channels 2cp / 2cp+1 for interior voxels [0..3] and for the
    // left / right edge voxels [4..11] (zero outside the volume: padding must stay 0 — max(0*x + 0, floor) = 0 for both floors, no select)
    float fcm[2][12];
    f32x4 fsc[2];                   // the row's (scale, shift) pairs: read from the LDS table (prep 0) or, D2, loaded with the halo
    float wt0 = 0.0f, wt1 = 0.0f;   // the output-pair write in flight: the two channels' transformed value, then its lo remainders
    unsigned whv = 0;               // ... and the packed hi halves
    auto halo_addr = [&](const PpPeriod &s, auto ROLEc) __attribute__((always_inline)) {
        constexpr int role = decltype(ROLEc)::value;
        const int gh = s.h0 - 1 + shl;
        int gd = s.d0 - 1 + sdl, fn = s.n;
        if constexpr (D2) { fn = s.n + (sdl >= 3 ? 1 : 0); gd = (sdl % 3) - 1; }   // slots [zero, n:d0, n:d1, zero, n+1:d0, n+1:d1]
        const bool in = (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && fn < N;
        o_row[role] = in ? (unsigned)((((long)fn * Ci + s.chunk * 16 + role * 8 + 2 * cp) * DHW + (long)gd * HW + gh * W + s.w0) * 4) : OOB;
        if constexpr (D2 && FUSE) t_row[role] = in ? (unsigned)((((long)fn * Ci + s.chunk * 16 + role * 8 + 2 * cp) * 2) * 4) : OOB;
    };
    // LD 0 / 1: voxels w0..w0+3 / w0+4..w0+7 of channel 2cp (one 16-byte load each), 2 / 3: of channel 2cp+1, 4 / 5: the edge voxels w0-1, w0+8
    // of channel 2cp / 2cp+1 (two 4-byte loads).  One statement each: a VMEM instruction holds the wave's issue while the CU's address unit
    // works off the other waves' — the step spaces them two MFMAs apart.
    auto halo_load = [&](const PpPeriod &s, auto ROLEc, auto LDc) __attribute__((always_inline)) {
        constexpr int role = decltype(ROLEc)::value, ld = decltype(LDc)::value;
        if (BT_ABL & 2) { asm volatile("" : "+v"(xa0[role]), "+v"(xb0[role]), "+v"(xa1[role]), "+v"(xb1[role]), "+v"(xl0[role]), "+v"(xr0[role]), "+v"(xl1[role]), "+v"(xr1[role])); return; }
        const bool in = o_row[role] != OOB;
        const unsigned o = o_row[role], o1 = in ? o + chan_stride : OOB;
        if constexpr (ld == 6) { if constexpr (D2 && FUSE) pp_buf_load_1x4(rsrc_aff, t_row[role], fsc[role]); }   // (out of the volume: zeros = padding stays 0)
        else if constexpr (ld == 0) pp_buf_load_1x4(rsrc, o, xa0[role]);
        else if constexpr (ld == 1) pp_buf_load_1x4(rsrc, in ? o + 16u : OOB, xb0[role]);
        else if constexpr (ld == 2) pp_buf_load_1x4(rsrc, o1, xa1[role]);
        else if constexpr (ld == 3) pp_buf_load_1x4(rsrc, in ? o1 + 16u : OOB, xb1[role]);
        else {
            const bool lft = in && s.w0 > 0, rgt = in && s.w0 + PP_TW < W;
            const unsigned ob = ld == 4 ? o : o1;
            pp_buf_load_2x1(rsrc, lft ? ob - 4u : OOB, rgt ? ob + 32u : OOB, ld == 4 ? xl0[role] : xl1[role], ld == 4 ? xr0[role] : xr1[role]);
        }
    };
    auto note = [&](float a, float b) {   // range diagnostic: one v_max3 (|a|, |b|, m) and one unordered compare per two values
        xmaxf_ = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)), xmaxf_);
        xnan_ |= __builtin_isunordered(a, b);
    };
    // PREP 0: read the row's (scale, shift) pairs; 1 / 2: fold the operand scale S (a power of two: (x m + a) S == x (m S) + a S and
    // max(., 0) S == max(. S, 0) bit for bit) and the padding masks into them
    auto halo_prep = [&](const PpPeriod &s, auto ROLEc, auto Kc) __attribute__((always_inline)) {
        constexpr int role = decltype(ROLEc)::value, k = decltype(Kc)::value;
        if (!fuse_in || (BT_ABL & 1)) return;
        if constexpr (k == 0) {
            if constexpr (!D2) fsc[role] = *reinterpret_cast<const f32x4 *>(aff + (s.chunk * 16 + role * 8 + 2 * cp) * 2);   // (scale, shift) of channel 2cp, then of 2cp+1
        } else {
            const int gh = s.h0 - 1 + shl;
            int gd = s.d0 - 1 + sdl, fn = s.n;
            if constexpr (D2) { fn = s.n + (sdl >= 3 ? 1 : 0); gd = (sdl % 3) - 1; }
            const float mid = ((unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && fn < N) ? x_scale : 0.0f;
            if constexpr (k == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) fcm[role][i] = fsc[role][i] * mid;
            } else {
                const float lft = s.w0 > 0 ? mid : 0.0f, rgt = s.w0 + PP_TW < W ? mid : 0.0f;
#pragma unroll
                for (int i = 0; i < 4; ++i) { fcm[role][4 + i] = fsc[role][i] * lft; fcm[role][8 + i] = fsc[role][i] * rgt; }
            }
        }
    };
    // CONVERT chunk c (in place): 0-3: two voxels each of channel 2cp (xa[0,1], xa[2,3], xb[0,1], xb[2,3]); 4 / 5: the edge voxels of channel
    // 2cp / 2cp+1; 6-9: channel 2cp+1
    auto halo_convert = [&](auto ROLEc, auto Cc) __attribute__((always_inline)) {
        constexpr int role = decltype(ROLEc)::value, c = decltype(Cc)::value;
        if (BT_ABL & 1) return;
        const float rf = relu_floor * x_scale;
        if constexpr (c == 4 || c == 5) {
            float &l = c == 4 ? xl0[role] : xl1[role], &r = c == 4 ? xr0[role] : xr1[role];
            constexpr int ch = c - 4;
            if (fuse_in) {
                l = fmaxf(l * fcm[role][4 + 2 * ch] + fcm[role][5 + 2 * ch], rf);
                r = fmaxf(r * fcm[role][8 + 2 * ch] + fcm[role][9 + 2 * ch], rf);
            } else { l *= x_scale; r *= x_scale; }
            note(l, r);
        } else {
            constexpr int ch = c >= 6 ? 1 : 0, qd = c >= 6 ? c - 6 : c;
            f32x4 &xv = ch == 0 ? (qd < 2 ? xa0[role] : xb0[role]) : (qd < 2 ? xa1[role] : xb1[role]);
            constexpr int e = 2 * (qd & 1);
            if (fuse_in) {
                xv[e] = fmaxf(xv[e] * fcm[role][2 * ch] + fcm[role][2 * ch + 1], rf);
                xv[e + 1] = fmaxf(xv[e + 1] * fcm[role][2 * ch] + fcm[role][2 * ch + 1], rf);
            } else { xv[e] *= x_scale; xv[e + 1] *= x_scale; }
            note(xv[e], xv[e + 1]);
        }
    };
    // WRITE stage k = 3 pp + st of output pair q: Winograd position pp of the row — st 0: the F(2,3) input transform (fp32, after the scale) of
    // the two channels; st 1: hi = rne_f16(t), t - hi as ONE v_fma_mix_f32 each; st 2: lo = rne_f16(t - hi), two half2 stores into the
    // role's buffer
    auto halo_write = [&](auto ROLEc, auto Qc, auto Kc) __attribute__((always_inline)) {
        constexpr int role = decltype(ROLEc)::value, q = decltype(Qc)::value, pp = decltype(Kc)::value / 3, st = decltype(Kc)::value % 3;
        if (BT_ABL & 1) { asm volatile("" :: "v"(xa0[role]), "v"(xb0[role]), "v"(xa1[role]), "v"(xb1[role]), "v"(xl0[role]), "v"(xr0[role]), "v"(xl1[role]), "v"(xr1[role])); return; }
        if constexpr (st == 0) {
            auto v0 = [&](int i) -> float { return i == 0 ? xl0[role] : i <= 4 ? xa0[role][i - 1] : i <= 8 ? xb0[role][i - 5] : xr0[role]; };   // (i folds: q is a constant)
            auto v1 = [&](int i) -> float { return i == 0 ? xl1[role] : i <= 4 ? xa1[role][i - 1] : i <= 8 ? xb1[role][i - 5] : xr1[role]; };
            if constexpr (pp == 0) { wt0 = v0(2 * q) - v0(2 * q + 2); wt1 = v1(2 * q) - v1(2 * q + 2); }
            else if constexpr (pp == 1) { wt0 = v0(2 * q + 1) + v0(2 * q + 2); wt1 = v1(2 * q + 1) + v1(2 * q + 2); }
            else if constexpr (pp == 2) { wt0 = v0(2 * q + 2) - v0(2 * q + 1); wt1 = v1(2 * q + 2) - v1(2 * q + 1); }
            else { wt0 = v0(2 * q + 1) - v0(2 * q + 3); wt1 = v1(2 * q + 1) - v1(2 * q + 3); }
        } else if constexpr (st == 1) {
            whv = pp_cvt_pk(wt0, wt1);
            wt0 = pp_sub_lo(whv, wt0);
            wt1 = pp_sub_hi(whv, wt1);
        } else {
            unsigned char *const xw = smem + PP_LDS_X + role * PP_XBUF_B + srow * PP_XROW_B + cp * 4;
            const unsigned lv = pp_cvt_pk(wt0, wt1);
            *reinterpret_cast<unsigned *>(xw + pp * PP_XPOS_B + q * PP_XPAIR_B) = whv;
            *reinterpret_cast<unsigned *>(xw + PP_XPART_B + pp * PP_XPOS_B + q * PP_XPAIR_B) = lv;
        }
    };
    auto load_aff = [&](int n, int first, int stride) {
        for (int i = first; i < Ci * 2; i += stride) aff[i] = in_affine[(size_t)n * Ci * 2 + i];
    };
    // the same table for another frame, as LDS-DMA (ONE wave, up to three pieces of 1 KiB; lanes beyond the table are masked off)
    const unsigned aff_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem + PP_LDS_AFF;
    auto dma_aff = [&](int n) __attribute__((always_inline)) {
        const unsigned char *const src = reinterpret_cast<const unsigned char *>(in_affine + (size_t)n * Ci * 2);
        const int bytes = Ci * 8;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i * 1024 < bytes && lane * 16 + i * 1024 < bytes)
                asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"((unsigned)(lane * 16 + i * 1024)), "s"(src), "s"(aff_lds + i * 1024) : "memory");
    };

    // ---- weight stream: piece i of a wave covers LDS bytes [(p + 4 i) KiB, +1 KiB) of the slab image (the role-split kernel's source
    // mapping: a lane's 16 bytes lie in k-group block (part, position, kg) = o / 1536; k-group 1 reads item 2s+1 = one slab further in the
    // pack, except at step 4 where item 9 is (tap 0, k-group 1))
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    unsigned dsrc[BT_PIECES], dsrc4[BT_PIECES];
#pragma unroll
    for (int i = 0; i < BT_PIECES; ++i) {
        const unsigned o = (unsigned)(p + 4 * i) * 1024u + (unsigned)lane * 16u;
        const unsigned blk = o / PP_KGBLK_B;
        const bool kg1 = (blk & 1u) != 0;
        const unsigned in_slab = (blk >> 1) * (2u * PP_KGBLK_B) + (o - blk * PP_KGBLK_B);
        dsrc[i] = in_slab + (kg1 ? (unsigned)PP_SLAB_B : 0u);
        dsrc4[i] = in_slab + (kg1 ? (unsigned)PP_KGBLK_B : 8u * PP_SLAB_B);
    }
    const unsigned char *const wbytes = reinterpret_cast<const unsigned char *>(wslabs) + (size_t)cot * nchunks * 9 * PP_SLAB_B;
    auto wchunk = [&](int chunk) -> const unsigned char * { return wbytes + (size_t)chunk * 9 * PP_SLAB_B; };
    auto dma_piece = [&](auto SQc, auto Kc, const unsigned char *base) __attribute__((always_inline)) {   // piece K (0..5) of slab SQ (0..8) of the period at `base`
        constexpr int sq = decltype(SQc)::value, k = decltype(Kc)::value;
        constexpr unsigned V0 = sq == 4 ? 0u : ((2 * sq) % 9) * PP_SLAB_B + ((2 * sq) / 9) * PP_KGBLK_B;   // (tap, k-group) of item 2 sq
        constexpr unsigned slot = sq % PP_R;
        if (BT_ABL & 4) return;
        bt_dma1(base + V0, sq == 4 ? dsrc4[k] : dsrc[k], lds0 + slot * PP_SLAB_B + (unsigned)(p + 4 * k) * 1024u);
    };
    auto dma_slab = [&](auto SQc, const unsigned char *base) __attribute__((always_inline)) {
        bt_for<BT_PIECES>([&](auto K) { dma_piece(SQc, K, base); });
    };

    // fragment bases (bytes).  X: the two k-groups of a step read items 2s and 2s+1 — one halo row apart (+16 B), eight rows apart (tap
    // (kd,2) -> (kd+1,0): +128 B), or (step 4) the last tap of buffer 0 and the first of buffer 1: three per-lane bases, the step's own
    // offset and the plane (t * 160 B) are immediates
    const unsigned a_off = (unsigned)(((p * 2 + kgl) * PP_COT + j) * 16);
    const unsigned b_lane = (unsigned)(PP_LDS_X + p * PP_XPOS_B + (j & 3) * PP_XPAIR_B + (j >> 2) * PP_XROW_B);
    const unsigned b_row = b_lane + (unsigned)kgl * PP_XROW_B, b_plane = b_lane + (unsigned)kgl * (8u * PP_XROW_B),
                   b_buf = b_lane + (unsigned)kgl * (unsigned)(PP_XBUF_B - pp_rowoff(8) * PP_XROW_B);
    const unsigned d2_k1 = 0u - (unsigned)(kgl * 3 * PP_HH * PP_XROW_B), d2_k0 = 0u - (unsigned)((1 - kgl) * 3 * PP_HH * PP_XROW_B);
    half8 ah[3], al[3], bh[4], bl[4];
    auto ld_a = [&](auto SPc, auto Mc, bool lo) __attribute__((always_inline)) -> half8 {   // weight fragment m of step sp
        constexpr int sp = decltype(SPc)::value, m = decltype(Mc)::value;
        if (BT_ABL & 32) { half8 z = {}; asm volatile("" : "+v"(z)); return z; }
        const unsigned char *const wsl = smem + (sp % PP_R) * PP_SLAB_B + a_off + m * 512;
        return *reinterpret_cast<const half8 *>(wsl + (lo ? PP_WPART_B : 0));
    };
    auto ld_b = [&](auto SPc, auto Tc, bool lo) __attribute__((always_inline)) -> half8 {   // X fragment of plane t of step sp
        constexpr int sp = decltype(SPc)::value, t = decltype(Tc)::value;
        if (BT_ABL & 32) { half8 z = {}; asm volatile("" : "+v"(z)); return z; }
        constexpr int I0 = 2 * sp, I1 = 2 * sp + 1;
        constexpr unsigned off0 = (I0 / 9) * PP_XBUF_B + pp_rowoff(I0 % 9) * PP_XROW_B;
        constexpr unsigned off1 = (I1 / 9) * PP_XBUF_B + pp_rowoff(I1 % 9) * PP_XROW_B;
        static_assert(off1 - off0 == PP_XROW_B || off1 - off0 == 8 * PP_XROW_B || off1 - off0 == PP_XBUF_B - pp_rowoff(8) * PP_XROW_B, "k-group distance");
        constexpr int PLANE = PP_HH * PP_XROW_B;
        constexpr int tb = D2 ? (t < 2 ? t : t + 1) : t;                        // first halo-plane slot of plane t
        constexpr bool z0 = D2 && t == 3 && (I0 % 9) / 3 == 2, z1 = D2 && t == 3 && (I1 % 9) / 3 == 2;   // k-group 0 / 1 would read slot 6: slot 3 instead
        const unsigned base = (off1 - off0 == PP_XROW_B ? b_row : off1 - off0 == 8 * PP_XROW_B ? b_plane : b_buf);
        const unsigned corr = z0 == z1 ? 0u : z1 ? d2_k1 : d2_k0;             // (per lane only where the two k-groups differ)
        const unsigned char *const xb = smem + (base + corr) + (int)(off0 + tb * PLANE) - (z0 && z1 ? 3 * PLANE : 0);
        return *reinterpret_cast<const half8 *>(xb + (lo ? PP_XPART_B : 0));
    };

    // ---- prologue: slabs 0, 1, 2 in flight, both halves of period 0 staged, the first step's Wlo / Xhi fragments loaded ---------------
    PpPeriod cur = period_at(j_first, c_begin), nxt = cur;
    bool nxt_ok = per_total > 1;
    if (nxt_ok) nxt = period_next(cur);
    if (tid < PP_COT) reinterpret_cast<float *>(smem + PP_LDS_BIAS)[tid] = (gridDim.z == 1 && bias) ? bias[cot * PP_COT + tid] : 0.0f;
    if (fuse_in && !D2) {
        load_aff(cur.n, tid, 256);
        aff_n = cur.n;
        lds_barrier();
    }
    dma_slab(std::integral_constant<int, 0>{}, wchunk(cur.chunk));
    dma_slab(std::integral_constant<int, 1>{}, wchunk(cur.chunk));
    dma_slab(std::integral_constant<int, 2>{}, wchunk(cur.chunk));
    bt_for<2>([&](auto R) {
        halo_addr(cur, R);
        bt_for<7>([&](auto L) { halo_load(cur, R, L); });
    });
    lds_dma_wait<0>();
    __builtin_amdgcn_sched_barrier(0);
    bt_for<2>([&](auto R) {
        bt_for<3>([&](auto K) { halo_prep(cur, R, K); });
        bt_for<10>([&](auto C) { halo_convert(R, C); });
        bt_for<4>([&](auto Q) { bt_for<12>([&](auto K) { halo_write(R, Q, K); }); });
    });
    lds_barrier();
    bt_for<3>([&](auto M) { al[decltype(M)::value] = ld_a(std::integral_constant<int, 0>{}, M, true); });
    bt_for<4>([&](auto T) { bh[decltype(T)::value] = ld_b(std::integral_constant<int, 0>{}, T, false); });

    const float unscale = whdr[0] * x_unscale;
    const int co0 = cot * PP_COT;
    const bool direct = gridDim.z == 1;
    int gp = 0;           // period being multiplied
    int epi_stores = 0;   // global stores of the last epilogue, younger than the slabs the next tile's first two waits ask for
#ifdef MPHIP_BT_PROFILE
    const unsigned long long prof_t1 = bt_memtime();
#endif

    for (int tj = j_first; tj < ntiles; tj += (int)gridDim.x) {
        const int en = cur.n, ed0 = cur.d0, eh0 = cur.h0, ew0 = cur.w0, etile = tile_at(tj);
        int tz = 0;
        asm volatile("" : "+v"(tz));  // opaque 0, new per tile: keeps the epilogue's per-channel address math / bias loads out of the tile loop
        f32x16 acc[3][4];
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.0f;

        for (int c = c_begin; c < c_end; ++c, ++gp) {
            // here: cur = the period being multiplied, nxt = its successor (if nxt_ok)
            const unsigned char *const wcur = wchunk(cur.chunk);
            const unsigned char *const wnxt = wchunk(nxt.chunk);
            const bool reload_aff = fuse_in && !D2 && nxt_ok && nxt.n != aff_n;   // (uniform)
            auto step = [&](auto SPc) __attribute__((always_inline)) {
                constexpr int sp = decltype(SPc)::value;
                constexpr int sn = (sp + 1) % 9;                       // the next step (of this or the next period): its fragments are prefetched
                using SP = std::integral_constant<int, sp>;
                using SN = std::integral_constant<int, sn>;
                using R0 = std::integral_constant<int, 0>;
                using R1 = std::integral_constant<int, 1>;
                // ---- the step's staging work as a list of micro-operations mop(0..16), hung one each behind the MFMAs of P2 (slots 3-11)
                // and P3 (slots 4-11).  Role 0 stages the even half of `nxt` into buffer 0: loads in step 0, table prep in 2, conversion in 3
                // (its loads were retired by the wait of step 3), output pairs 0-3 written in steps 4-7.  Role 1 stages the odd half into
                // buffer 1: loads in step 3, prep in 5, conversion in 6-7, pair 0 written in step 8 and pairs 1-3 in steps 0-2 of the period
                // the data belongs to (there it is `cur`).  Nothing of this is conditional — a branch per piece costs the single wave two issue
                // slots: in the last period `nxt` is a stale copy of a valid period (redundant loads, writes into buffers nobody reads again),
                // and period 0's pairs 1-3 are re-written from the registers the prologue converted (identical bytes).
                auto mop = [&](auto Kc) __attribute__((always_inline)) {
                    constexpr int k = decltype(Kc)::value;
                    if constexpr (sp == 3) {            // role 0's conversion (10 chunks)
                        if constexpr (k < 10) halo_convert(R0{}, std::integral_constant<int, (k < 10 ? k : 0)>{});
                    } else if constexpr (k < 12) {      // the step's output pair
                        if constexpr (sp >= 4 && sp < 8) halo_write(R0{}, std::integral_constant<int, (sp >= 4 && sp < 8 ? sp - 4 : 0)>{}, Kc);
                        else if constexpr (sp == 8) halo_write(R1{}, std::integral_constant<int, 0>{}, Kc);
                        else if constexpr (sp < 3) halo_write(R1{}, std::integral_constant<int, (sp < 3 ? sp + 1 : 0)>{}, Kc);
                    } else {
                        constexpr int e = k - 12;       // 0..4
                        if constexpr (sp == 2) { if constexpr (e < 3) halo_prep(nxt, R0{}, std::integral_constant<int, (e < 3 ? e : 0)>{}); }
                        else if constexpr (sp == 5) { if constexpr (e < 3) halo_prep(nxt, R1{}, std::integral_constant<int, (e < 3 ? e : 0)>{}); }
                        else if constexpr (sp == 6) halo_convert(R1{}, std::integral_constant<int, e>{});
                        else if constexpr (sp == 7) halo_convert(R1{}, std::integral_constant<int, 5 + e>{});
                    }
                };
                // ---- the step's vector-memory operations, ONE per slot and at least two MFMAs apart (the four waves reach the same slot
                // together, and the CU's address unit needs ~16 cycles per 16-byte-per-lane instruction: a wave whose VMEM instruction waits
                // for it cannot issue its MFMAs either).  seg 2 = P2, 3 = P3.  Slab sp+3's six pieces: P2 slots 3, 7, 11, P3 slots 5, 8, 11
                // (past the stream's end they re-read a valid slab into a slot nobody reads again: no branch).  The halo loads of role 0
                // (step 0) / role 1 (step 3): P2 slots 5, 9, P3 slots 4, 6, 9, 10.
                const unsigned char *const wsrc = (sp + 3 < 9) ? wcur : wnxt;
                auto vm = [&](auto SEGc, auto Ic) __attribute__((always_inline)) {
                    constexpr int seg = decltype(SEGc)::value, i = decltype(Ic)::value;
                    constexpr int piece = seg == 2 ? (i == 3 ? 0 : i == 7 ? 1 : i == 11 ? 2 : -1) : (i == 5 ? 3 : i == 8 ? 4 : i == 11 ? 5 : -1);
                    constexpr int ld = seg == 2 ? (i == 5 ? 0 : i == 9 ? 1 : -1) : (i == 4 ? 2 : i == 6 ? 3 : i == 9 ? 4 : i == 10 ? 5 : (i == 7 && D2 && FUSE) ? 6 : -1);
                    if constexpr (piece >= 0) dma_piece(std::integral_constant<int, (sp + 3) % 9>{}, std::integral_constant<int, (piece >= 0 ? piece : 0)>{}, wsrc);
                    if constexpr (ld >= 0 && sp == 0) halo_load(nxt, R0{}, std::integral_constant<int, (ld >= 0 ? ld : 0)>{});
                    if constexpr (ld >= 0 && sp == 3) halo_load(nxt, R1{}, std::integral_constant<int, (ld >= 0 ? ld : 0)>{});
                };
                // ---------------- P1(sp) = Wlo x Xhi; behind its first MFMAs: Whi(sp), Xlo(sp) — nothing behind the last five, so that no LDS
                // operation is young when the wave reaches the barrier
                bt_for<12>([&](auto I) {
                    constexpr int i = decltype(I)::value, m = i / 4, t = i % 4;
                    if (BT_ABL & 16) asm volatile("" ::"v"(al[m]), "v"(bh[t])); else acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh[t], acc[m][t], 0, 0, 0);
                    if constexpr (i < 3) ah[i] = ld_a(SP{}, I, false);
                    else if constexpr (i < 7) bl[i - 3] = ld_b(SP{}, std::integral_constant<int, (i >= 3 && i < 7 ? i - 3 : 0)>{}, true);
                    else if constexpr (i == 8 && sp == 0) halo_addr(nxt, R0{});
                    else if constexpr (i == 8 && sp == 3) halo_addr(nxt, R1{});
                    __builtin_amdgcn_sched_barrier(0);
                });
                // ---------------- the step's wait + barrier: slab sp+1 (this wave's pieces, issued two steps ago) has landed; slab sp is dead.
                // Younger than those pieces and left in flight: everything the previous step issued (its six pieces, its halo loads) and — in a
                // tile's first two steps — the epilogue's stores.  (vmcnt's immediate is a constant; an under-estimate would only wait longer)
                constexpr int HL1 = 8 + (D2 && FUSE ? 1 : 0);                          // (+ the row's table values in the two-frame mode)
                constexpr int hl_[9] = {HL1, 0, 0, HL1, 0, 0, 0, 0, 0};               // halo load instructions of a step: roles 0 (step 0) and 1 (step 3)
                constexpr int HP = hl_[(sp + 8) % 9];
                if (BT_ABL & 64) {}
                else if (sp < 2 && epi_stores == 48) lds_dma_wait<BT_PIECES + 48 + HP>();
                else if (sp < 2 && epi_stores == 24) lds_dma_wait<BT_PIECES + 24 + HP>();
                else if (sp < 2 && epi_stores == 12) lds_dma_wait<BT_PIECES + 12 + HP>();
                else lds_dma_wait<BT_PIECES + HP>();
                if (sp == 1) epi_stores = 0;
                __builtin_amdgcn_sched_barrier(0);
                if (!(BT_ABL & 64)) lds_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---------------- P2(sp) = Whi x Xhi; behind its MFMAs: Wlo(sp+1), the fused-GroupNorm table of the next period's frame (step 0,
                // one wave: its last reader was step 5's prep, the next one is step 2's; issued ahead of the step's pieces, so the wait of step
                // 2 retires it), the six pieces of slab sp+3 — into the slot of slab sp — and mop 0-8
                bt_for<12>([&](auto I) {
                    constexpr int i = decltype(I)::value, m = i / 4, t = i % 4;
                    if (BT_ABL & 16) asm volatile("" ::"v"(ah[m]), "v"(bh[t])); else acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh[t], acc[m][t], 0, 0, 0);
                    if constexpr (i < 3) {
                        al[i] = ld_a(SN{}, I, true);
                        if constexpr (sp == 0 && i == 2) { if (reload_aff && p == 0) dma_aff(nxt.n); }
                    } else {
                        vm(std::integral_constant<int, 2>{}, I);
                        mop(std::integral_constant<int, (i >= 3 ? i - 3 : 0)>{});
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                // ---------------- P3(sp) = Whi x Xlo; behind its MFMAs: Xhi(sp+1) and mop 9-16
                bt_for<12>([&](auto I) {
                    constexpr int i = decltype(I)::value, m = i / 4, t = i % 4;
                    if (BT_ABL & 16) asm volatile("" ::"v"(ah[m]), "v"(bl[t])); else acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl[t], acc[m][t], 0, 0, 0);
                    if constexpr (i < 4) bh[i] = ld_b(SN{}, I, false);
                    else {
                        vm(std::integral_constant<int, 3>{}, I);
                        mop(std::integral_constant<int, (i >= 4 ? 9 + i - 4 : 0)>{});
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            };
            step(std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{});
            step(std::integral_constant<int, 4>{});
            step(std::integral_constant<int, 5>{});
            step(std::integral_constant<int, 6>{});
            step(std::integral_constant<int, 7>{});
            step(std::integral_constant<int, 8>{});
            if (reload_aff) aff_n = nxt.n;
            cur = nxt;
            nxt_ok = gp + 2 < per_total;
            if (nxt_ok) nxt = period_next(cur);
        }

        // ---- output transform + epilogue: six rounds (32-channel row tile x plane pair) through the exchange region; the four positions of a
        // pair live in the four waves: each wave parks the units the others finish and finishes its own.  (X buffer 1 is NOT a second exchange
        // region here: role 1 has already written the next period's first output pair into it.)  Same per-element arithmetic, GroupNorm
        // partials and store pattern as the role-split kernel.
#ifdef MPHIP_BT_PROFILE
        const unsigned long long prof_e0 = bt_memtime();
#endif
        if (BT_ABL & 8) {
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int t = 0; t < 4; ++t) asm volatile("" ::"a"(acc[m][t]));
            continue;
        }
        const int gn_rows = tiles_total * 2;   // channel-major [Co][tile * 2 + plane pair][2] (the finalize kernel reads rows of it)
        if (gn_part && etile == 0 && tid == 0) gn_part[(size_t)gn_rows * Co * 2] = unscale;   // (behind the partials)
        const bool odd = (lane & 1) != 0;
        // (this lane's QUAD of voxels: lanes 2k / 2k+1 store the 4 voxels 4k..4k+3 of a row, for different channels)
        // (D2: plane pair = frame: pair 1 is frame en + 1, d = 0, 1 — absent when N is odd and this is the last tile)
        float *const ybase = (direct ? y : y + (size_t)blockIdx.z * N * Co * DHW) + ((size_t)en * Co + co0) * DHW + (size_t)ed0 * HW + (size_t)eh0 * W + ew0;
        const size_t pair_stride = D2 ? (size_t)Co * DHW : (size_t)2 * HW;
        const bool pair1_ok = !D2 || en + 1 < N;
        const unsigned yoff = (unsigned)(((8 * p + 4 * kgl + (odd ? 2 : 0)) * DHW + (j >> 2) * W + 2 * (j & 2)) * 4);
        auto rounds = [&](auto Pc) __attribute__((always_inline)) {
            constexpr int P = decltype(Pc)::value;
            float *const Ex = reinterpret_cast<float *>(smem + PP_LDS_EX);   // [position][slot 0..5][lane][4]
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                // (the accumulators live in AGPRs; without this pin hipcc hoists the v_accvgpr_read of ALL 192 above the switch over the
                //  wave's position and spills ~100 registers around every tile's epilogue — a round needs its own 64 only)
#pragma unroll
                for (int t = 0; t < 4; ++t) asm volatile("" : "+a"(acc[m][t]));
                const f32x4 bv = *reinterpret_cast<const f32x4 *>(smem + PP_LDS_BIAS + (m * 32 + 8 * P + 4 * kgl) * 4);
#pragma unroll
                for (int pair = 0; pair < 2; ++pair) {
                    // park the units other waves finish: unit u = accumulator registers 4u..4u+3 of a column tile; wave P keeps unit P
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (u != P) {
                                const int slot = tt * 3 + (u - (u > P ? 1 : 0));
                                const f32x16 &a = acc[m][2 * pair + tt];
                                const f32x4 v = {a[4 * u], a[4 * u + 1], a[4 * u + 2], a[4 * u + 3]};
                                *reinterpret_cast<f32x4 *>(Ex + ((P * 6 + slot) * 64 + lane) * 4) = v;
                            }
                    lds_barrier();
                    float ssum[4] = {0.0f, 0.0f, 0.0f, 0.0f}, qsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        f32x4 M[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            if (q != P) {
                                const int slot = tt * 3 + (P - (P > q ? 1 : 0));
                                M[q] = *reinterpret_cast<const f32x4 *>(Ex + ((q * 6 + slot) * 64 + lane) * 4);
                            } else {
#pragma unroll
                                for (int i = 0; i < 4; ++i) M[q][i] = acc[m][2 * pair + tt][4 * q + i];
                            }
                        }
                        float y0[4], y1[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float r0 = (M[0][i] + M[1][i]) + M[2][i];
                            const float r1 = (M[1][i] - M[2][i]) - M[3][i];
                            if (gn_part) {   // (uniform)
                                ssum[i] += r0 + r1;
                                qsum[i] = __builtin_fmaf(r0, r0, qsum[i]);
                                qsum[i] = __builtin_fmaf(r1, r1, qsum[i]);
                            }
                            y0[i] = r0 * unscale + bv[i];
                            y1[i] = r1 * unscale + bv[i];
                        }
                        // 16-byte stores: a lane holds one output pair (2 voxels) of 4 channels; lanes 2k / 2k+1 hold neighbouring pairs of a row
                        // and trade halves (quad_perm [1,0,3,2]): the even lane ends up with 4 consecutive voxels of channels 0-1, the odd lane
                        // with those of channels 2-3
#define BT_SWAP(v_) __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v_), 0xB1, 0xf, 0xf, false))
                        const float g0 = BT_SWAP(odd ? y0[0] : y0[2]), g1 = BT_SWAP(odd ? y1[0] : y1[2]);
                        const float g2 = BT_SWAP(odd ? y0[1] : y0[3]), g3 = BT_SWAP(odd ? y1[1] : y1[3]);
#undef BT_SWAP
                        const f32x4 va = {odd ? g0 : y0[0], odd ? g1 : y1[0], odd ? y0[2] : g0, odd ? y1[2] : g1};
                        const f32x4 vb = {odd ? g2 : y0[1], odd ? g3 : y1[1], odd ? y0[3] : g2, odd ? y1[3] : g3};
                        unsigned char *const dq = reinterpret_cast<unsigned char *>(ybase + (size_t)(m * 32 + tz) * DHW + pair * pair_stride + (size_t)tt * HW);   // (uniform)
                        if (pair == 0 || pair1_ok) {
                            *reinterpret_cast<f32x4 *>(dq + yoff) = va;
                            *reinterpret_cast<f32x4 *>(dq + (size_t)DHW * 4 + yoff) = vb;
                        }
                    }
                    if (gn_part) {
                        // per-channel (sum, sum of squares) of the RAW transformed accumulators over this wave's 2 x 64 voxels of the channel and
                        // plane pair: the 32 lanes of a half-wave hold one channel's columns (the finalize kernel applies unscale and the bias)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
#define BT_ROW_ADD(v_, ctrl_) v_ += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v_), ctrl_, 0xf, 0xf, false));
                            BT_ROW_ADD(ssum[i], 0x128) BT_ROW_ADD(qsum[i], 0x128)   // row_ror:8, :4, :2, :1 -> every lane of a 16-lane row: the row's sum
                            BT_ROW_ADD(ssum[i], 0x124) BT_ROW_ADD(qsum[i], 0x124)
                            BT_ROW_ADD(ssum[i], 0x122) BT_ROW_ADD(qsum[i], 0x122)
                            BT_ROW_ADD(ssum[i], 0x121) BT_ROW_ADD(qsum[i], 0x121)
#undef BT_ROW_ADD
                            // rows 1 and 3 add the totals of rows 0 and 2 (row_bcast:15, row mask 0b1010): lanes 16-31 / 48-63 hold a half-wave's sum
                            ssum[i] += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(ssum[i]), 0x142, 0xa, 0xf, false));
                            qsum[i] += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(qsum[i]), 0x142, 0xa, 0xf, false));
                        }
                        if (j == 31) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int co = co0 + m * 32 + 8 * P + 4 * kgl + i + tz;
                                *reinterpret_cast<float2 *>(gn_part + ((size_t)co * gn_rows + (size_t)etile * 2 + pair) * 2) = make_float2(ssum[i], qsum[i]);
                            }
                        }
                    }
                    lds_barrier();   // the region is rewritten by the next round
                }
            }
        };
        switch (p) {   // (wave-uniform)
        case 0: rounds(std::integral_constant<int, 0>{}); break;
        case 1: rounds(std::integral_constant<int, 1>{}); break;
        case 2: rounds(std::integral_constant<int, 2>{}); break;
        default: rounds(std::integral_constant<int, 3>{}); break;
        }
        epi_stores = gn_part ? 48 : pair1_ok ? 24 : 12;   // (an odd batch's last tile stores one frame only: counted, an over-estimate would end a wait early)
#ifdef MPHIP_BT_PROFILE
        prof_epi += bt_memtime() - prof_e0;
#endif
    }

    // operands outside the f16 range (non-finite inputs, or finite ones beyond a wrong caller-supplied descriptor) are not clamped — they
    // propagate as Inf / NaN — but they are counted, per thread that saw any
    const bool sat_ = xnan_ || xmaxf_ > 0.5f * F16_CLAMP;
    if (__builtin_amdgcn_ballot_w64(sat_) != 0) {  // never taken in normal operation
        unsigned tot = sat_;
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) tot += __shfl_xor(tot, sft, 64);
        if (lane == 0) atomicAdd(&g_f16x3_wino_bt_saturated, (unsigned long long)tot);
    }
#ifdef MPHIP_BT_PROFILE
    if (lane == 0) {
        const unsigned long long t = bt_memtime();
        atomicAdd(&g_bt_prof[0], t - prof_t0);
        atomicAdd(&g_bt_prof[1], prof_epi);
        atomicAdd(&g_bt_prof[2], 1ull);
        atomicAdd(&g_bt_prof[3], prof_t1 - prof_t0);
    }
#endif
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
int f16x3_wino_bt_saturation(unsigned long long *count, int reset) {
    if (hipMemcpyFromSymbol(count, HIP_SYMBOL(g_f16x3_wino_bt_saturated), sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        const unsigned long long z = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_f16x3_wino_bt_saturated), &z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}

void f16x3_wino_bt_launch(dim3 grid, hipStream_t s, hipEvent_t t0, hipEvent_t t1, const float *x, const _Float16 *slabs, const float *hdr,
                          const float *bias, float *dst, int N, int Ci, int Co, int D, int H, int W, int cps, unsigned xb,
                          const float *in_affine, int in_relu, const float *x_range, int tiles, int xcd_on, const int *tile_list,
                          float *gn_part) {
#define BT_LAUNCH(F_, D_)                                                                                                                      \
    {                                                                                                                                          \
        if (t0 && t1)                                                                                                                          \
            hipExtLaunchKernelGGL((conv3d_k3_f16x3_wino_bt_kernel<F_, D_>), grid, dim3(256), 0, s, t0, t1, 0, x, slabs, hdr, bias, dst, N, Ci, Co, D, H, \
                                  W, cps, xb, in_affine, in_relu, x_range, tiles, xcd_on, tile_list, gn_part);                                 \
        else                                                                                                                                   \
            hipLaunchKernelGGL((conv3d_k3_f16x3_wino_bt_kernel<F_, D_>), grid, dim3(256), 0, s, x, slabs, hdr, bias, dst, N, Ci, Co, D, H, W, cps, xb,   \
                               in_affine, in_relu, x_range, tiles, xcd_on, tile_list, gn_part);                                                \
    }
    if (D == 2) { if (in_affine) BT_LAUNCH(true, true) else BT_LAUNCH(false, true) }
    else if (in_affine) BT_LAUNCH(true, false) else BT_LAUNCH(false, false)
#undef BT_LAUNCH
}

}  // namespace mphip

#ifdef MPHIP_BT_PROFILE
extern "C" int mphip_debug_wino_bt_profile(unsigned long long *out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(mphip::g_bt_prof), 64) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(mphip::g_bt_prof), z, 64) != hipSuccess) return -1;
    }
    return 0;
}
#endif
