// K4 fast mode, transformed domain, role-split schedule — the F(2,3) conv of conv3d_f16x3_wino.hip (same arithmetic, same packed
// weights, same tile) with the two waves of every SIMD working in OPPOSITE phases ("ping-pong").
//
// Why (r05): the lockstep kernel's own ablations put its non-matrix work at 0.37 ms and its MFMAs at 0.25 ms of a
// 0.56 ms launch — nearly serial: all eight waves read fragments, issue LDS-DMA, stage the halo and run their MFMAs in the SAME
// barrier-delimited interval, so the matrix pipe idles while every wave loads and the LDS idles while every wave multiplies
// (MFMA pipe 45 % busy).  cdna_hip_programming.md 5.5 (T3/T5) and MI355X_MICROARCH.md "Two waves per SIMD": what feeds the pipe on
// this part is a schedule in which one wave of a SIMD is in a pure-MFMA segment while its partner is in its load segment.
//
// Schedule.  Waves 0-3 ("team A": plane pair 0 of the tile) and 4-7 ("team B": plane pair 1) run the SAME program
//      LOAD(k) | barrier | MFMA(k) | barrier | LOAD(k+1) | ...
// but team B starts one barrier late, so A's MFMA(k) coincides with B's LOAD(k) and B's MFMA(k) with A's LOAD(k+1).  Wave w
// and w+4 share a SIMD: its matrix pipe always has exactly one wave issuing 18 back-to-back MFMAs (s_setprio 1) while the other
// does everything else: the fragment reads of its next step (single-buffered: the previous fragments are dead), its three
// LDS-DMA pieces, its slice of the halo staging, its vmcnt wait.
//
// K loop.  The X tile is staged 8 input channels at a time (30 KB per buffer, two buffers = what ONE 16-channel tile took), and
// a K = 16 MFMA step pairs two consecutive (8-channel chunk, (kd,kh) tap) ITEMS: k-group 0 of a step is item 2s, k-group 1 item
// 2s+1 of the flattened list (chunk-major, 9 taps per chunk).  A 16-channel "period" is 9 steps; step 4 reads both buffers.
// Buffer 0 is dead during steps 5-8 and buffer 1 during steps 0-3 of the next period: each buffer is rewritten in the window in
// which it is not read — double buffering without a second 62 KB tile, no staging bubble at chunk boundaries.  Team A stages
// the even chunks (k-group 0 of the pack's 16-channel chunk) during its LOAD phases of steps 5-8, one output pair per phase;
// team B the odd chunks during steps 0-3.  The weight slabs are the lockstep kernel's ([part][position][kg][co][8] per (chunk16,
// tap)): the pairing is applied by the LDS-DMA's per-lane SOURCE address (k-group block kg of step s comes from tap / k-group
// (item % 9, item / 9) of the pack) — no second pack.
//
// LDS (163 200 B): ring of 3 weight slabs (72 KB) | X, two 8-channel buffers (60 KB) | team A's output-transform exchange
// (24 KB; team B's is X buffer 1, dead at a tile's end) | fused-GroupNorm table (3 KB) | bias (384 B).
// Ring hazards (phases numbered globally, A: LOAD(k) = 2k, MFMA(k) = 2k+1; B: one later): slab k is read in phases 2k (A) and
// 2k+1 (B); a wave issues its pieces of slab k+2 in ITS LOAD(k) — into the slot of slab k-1, last read in phase 2k-1 — waits
// for them (vmcnt) at the start of ITS next LOAD phase and publishes them with that phase's barrier: A's pieces are visible
// from phase 2k+3, B's from 2k+4 = A's LOAD(k+2).  Every transfer has two phases (> 1100 cycles) to land.
#include <stdlib.h>

#include <type_traits>

#include <hip/hip_ext.h>

#include "mphip_ablate.h"
#include "mphip_conv.h"
#include "mphip_f16x3.h"
#include "mphip_wino_tile.h"

namespace mphip {


__device__ unsigned long long g_f16x3_wino_pp_saturated;

#ifdef MPHIP_PP_PROFILE
// dev instrumentation: wall time (shader cycles) of every phase by step-in-period, summed over waves.  A time stamp is taken at the START of a
// phase, i.e. right after a barrier.
// g_pp_prof[team * 32 + k]: k = 2 sp: LOAD(sp) incl. its barrier, 2 sp + 1: MFMA(sp) incl. its barrier (k = 17 also carries a tile's
// epilogue and the prologue), 20 waves
// k = 21..27: the epilogue: 21 = its first segment (slab pre-issue + exchange writes of round 0 + barrier), 22 = round 0 transform + stores + barrier,
// 23 / 24 and 25 / 26 = rounds 1 and 2, 27 = from the epilogue's last barrier to the next LOAD's stamp
__device__ unsigned long long g_pp_prof[64];
// (s_memtime as volatile asm WITH its wait: hipcc hoists __builtin_readcyclecounter() across a whole LOAD segment, and an un-waited asm result
//  lands in an SGPR pair the compiler may already have spilled and reused — for a pointer, as it happened.  Every stamp follows a barrier,
//  where lgkmcnt is 0 anyway: the wait costs the scalar-memory round trip, ~+10 % on a phase)
__device__ __forceinline__ unsigned long long pp_memtime() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define PPROF_DECL unsigned long long pt0_ = pp_memtime(), pt1_ = pt0_, pa_[18] = {}, pe_[7] = {}, pet_ = 0
#define PPROF_EPI(k) { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = pp_memtime(); if ((k) > 0) pe_[(k) - 1] += n_ - pet_; pet_ = n_; __builtin_amdgcn_sched_barrier(0); }
#define PPROF_STAMP(j) { __builtin_amdgcn_sched_barrier(0); pa_[((j) + 16) % 18] += pt1_ - pt0_; pt0_ = pt1_; pt1_ = pp_memtime(); __builtin_amdgcn_sched_barrier(0); }
#define PPROF_FLUSH { PPROF_STAMP(0) PPROF_STAMP(1) if ((threadIdx.x & 63) == 0) { for (int q_ = 0; q_ < 18; ++q_) atomicAdd(&g_pp_prof[team * 32 + q_], pa_[q_]); atomicAdd(&g_pp_prof[team * 32 + 20], 1ull); for (int q_ = 0; q_ < 7; ++q_) atomicAdd(&g_pp_prof[team * 32 + 21 + q_], pe_[q_]); } }
#else
#define PPROF_DECL
#define PPROF_EPI(k)
#define PPROF_STAMP(j)
#define PPROF_FLUSH
#endif

// Priority policy (dev A/B, -DPP_PRIO=n): 0 none; 1 s_setprio 1 around every MFMA segment; 2 static: team B (the younger half) at 1;
// 3 s_setprio 1 around every LOAD segment
#ifndef PP_PRIO
#define PP_PRIO 0
#endif
// Timing-only ablations (dev, wrong results; -DPP_ABL=mask): the bits are listed in mphip_ablate.h

// Three LDS-DMA pieces of 16 bytes per lane: global `base` (wave-uniform) + `off_i` (per lane) -> LDS byte address `lds_i` + 16 * lane.
// Hand-issued for the reason given at lds_dma16 (mphip_f16x3.h); M0 is written in the statement that reads it.  The statement opens
// with `s_nop 4`: `base` may have been reloaded from a spill lane by v_readlane just before, and a VALU-written SGPR needs 5 wait
// states before a vector-memory instruction reads it — hipcc pads nothing inside an asm string (cdna_hip_programming.md 5.7).
// (No "m0" clobber: hipcc refuses it — "reserved register, undefined behaviour".  What it would promise is checked on the disassembly by
//  tests/test_host.py: outside these statements the kernel contains no instruction that reads or writes M0.)
__device__ __forceinline__ void pp_dma3(const void *base, unsigned off0, unsigned off1, unsigned off2, unsigned lds0, unsigned lds1, unsigned lds2) {
    asm volatile("s_nop 4\n\t"
                 "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %3\n\t"
                 "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
                 "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3"
                 ::"v"(off0), "v"(off1), "v"(off2), "s"(base), "s"(lds0), "s"(lds1), "s"(lds2) : "memory");
}


// SINGLE: the reference's autocast(float16) policy for these convs (train.py:145,188) — ONE f16 product per multiply (the hi halves only:
// operands rounded to f16 in the transformed domain, fp32 accumulation), a third of the MFMAs; everything else unchanged.
template <bool SINGLE>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
conv3d_k3_f16x3_wino_pp_kernel(const float *__restrict__ x, const _Float16 *__restrict__ wslabs, const float *__restrict__ whdr,
                               const float *__restrict__ bias, float *__restrict__ y, int N, int Ci, int Co, int D, int H, int W,
                               int chunks_per_split, unsigned x_bytes, const float *__restrict__ in_affine, int in_relu,
                               const float *__restrict__ x_range, int tiles_total, int xcd_aware, const int *__restrict__ tile_list,
                               float *__restrict__ gn_part) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[PP_LDS_BYTES];
    float *const aff = reinterpret_cast<float *>(smem + PP_LDS_AFF);   // [Ci][2]: (scale, shift) of the fused input GroupNorm

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = wave & 3, team_rt = wave >> 2;                // Winograd position; plane pair (waves w and w+4 share a SIMD)
    const int j = lane & 31, kgl = lane >> 5;
    const int HW = H * W, DHW = D * HW;
    const int ntiles = tile_list ? tile_list[0] : tiles_total;
    auto tile_at = [&](int jj) -> int { return tile_list ? tile_list[1 + jj] : jj; };
    const int j_first = (tile_list || !xcd_aware) ? (int)blockIdx.x : (int)xcd_remap(blockIdx.x, gridDim.x);
    if (j_first >= ntiles) return;   // (workgroup-uniform, before any barrier)
    float x_scale = 16.0f, x_unscale = 1.0f / 16.0f;
    if (x_range) range_scale_block(x_range, x_scale, x_unscale);

    const int tiles_w = W / PP_TW, tiles_h = H / PP_TH, tiles_d = D / PP_TD;
    const int cot = blockIdx.y;
    const int nchunks = Ci / 16;
    const int c_begin = blockIdx.z * chunks_per_split, c_end = min(nchunks, c_begin + chunks_per_split);
    if (c_begin >= c_end) return;
    const int nper = c_end - c_begin;
    const int nmine = (ntiles - j_first + (int)gridDim.x - 1) / (int)gridDim.x;   // tiles this workgroup walks
    const int per_total = nmine * nper;                                           // 16-channel periods it walks
    const int s_total = per_total * 9;                                            // slabs it consumes

    auto period_at = [&](int tj, int chunk) -> PpPeriod {
        PpPeriod r;
        int bid = tile_at(tj);
        const int tw = bid % tiles_w; bid /= tiles_w;
        const int th = bid % tiles_h; bid /= tiles_h;
        const int td = bid % tiles_d;
        r.n = bid / tiles_d;
        r.d0 = td * PP_TD; r.h0 = th * PP_TH; r.w0 = tw * PP_TW;
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

    // Everything below is instantiated ONCE PER TEAM (team = a compile-time constant): the two teams execute the same barrier sequence,
    // but no team-dependent branch is left inside — a team's staging registers, whose loads are ordered by hand, never meet the other
    // team's values at a control-flow join (where hipcc would be free to insert copies of registers that are still in flight).
    auto run = [&](auto TEAMc) __attribute__((always_inline)) {
    constexpr int team = decltype(TEAMc)::value;
    // ---- X staging: a team's thread = (channel pair cp of ITS 8-channel half of the period, halo row); 240 of a team's 256 threads
    // (the other 16 re-do row 59: same loads, same values, same LDS addresses — branch-free code)
    const int tt = tid & 255;
    const int cp = tt & 3, srow = min(tt >> 2, PP_ROWS - 1);
    const int sdl = srow / PP_HH, shl = srow % PP_HH;
    const bool fuse_in = in_affine != nullptr;   // workgroup-uniform
    const float relu_floor = in_relu ? 0.0f : -3.0e38f;
    int aff_n = -1;
    // ONE register set per thread for the row of its channel pair (w0-1 .. w0+8): loaded, normalised / scaled IN PLACE, transformed from —
    // the two teams run the same code half a period apart, so separate "raw" and "scaled" variables would both be live everywhere.
    // The eight loads of a unit go out over THREE LOAD phases (part 0: channel 2cp, 1: channel 2cp+1, 2: the four edge voxels): all 256 CUs
    // run this schedule in lockstep, and a whole unit in one phase is a chip-wide burst at the HBM rate (~10 B/clk/CU: a 4 us phase).
    f32x4 xa0, xb0, xa1, xb1;   // channels 2cp / 2cp+1 of the half: voxels w0..w0+3, w0+4..w0+7
    float xl0, xr0, xl1, xr1;   // ... w0-1, w0+8
    float xmaxf_ = 0.0f;        // max |scaled halo value| this thread staged (finite or Inf) ...
    bool xnan_ = false;         // ... and whether it saw a NaN (v_max drops them)
    unsigned o_row = OOB;       // byte offset of (n, channel 2cp of the half, row, w0) of the unit being loaded, or OOB (padding rows): kept from part 0
    auto halo_load = [&](const PpPeriod &s, auto PARTc) __attribute__((always_inline)) {
        constexpr int part = decltype(PARTc)::value;
        if (PP_ABL & 2) { asm volatile("" : "+v"(xa0), "+v"(xb0), "+v"(xa1), "+v"(xb1), "+v"(xl0), "+v"(xr0), "+v"(xl1), "+v"(xr1)); return; }
        if constexpr (part == 0) {
            const int gd = s.d0 - 1 + sdl, gh = s.h0 - 1 + shl;
            const bool in = (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H;
            o_row = in ? (unsigned)((((long)s.n * Ci + s.chunk * 16 + team * 8 + 2 * cp) * DHW + (long)gd * HW + gh * W + s.w0) * 4) : OOB;
        }
        const bool in = o_row != OOB;
        const unsigned o = o_row, o1 = in ? o + chan_stride : OOB;
        if constexpr (part == 0) {
            pp_buf_load_2x4(rsrc, o, in ? o + 16u : OOB, xa0, xb0);
        } else if constexpr (part == 1) {
            pp_buf_load_2x4(rsrc, o1, in ? o1 + 16u : OOB, xa1, xb1);
        } else {
            const bool lft = in && s.w0 > 0, rgt = in && s.w0 + PP_TW < W;
            pp_buf_load_4x1(rsrc, lft ? o - 4u : OOB, rgt ? o + 32u : OOB, lft ? o1 - 4u : OOB, rgt ? o1 + 32u : OOB, xl0, xr0, xl1, xr1);
        }
    };
    // (fused GroupNorm + ReLU) -> operand scale, in place; part 0: channel 2cp, 1: channel 2cp+1, 2: the four edge voxels — one part per LOAD
    // phase, each at least two phases after its loads (the counted waits in between have retired them).  The scale S is a power of two:
    // (x m + a) S == x (m S) + a S and max(., 0) S == max(. S, 0) bit for bit, so the fused path folds S into the table entries.
    auto halo_convert = [&](const PpPeriod &s, auto PARTc) __attribute__((always_inline)) {
        constexpr int part = decltype(PARTc)::value;
        if (PP_ABL & 1) { asm volatile("" :: "v"(xa0), "v"(xb0), "v"(xa1), "v"(xb1), "v"(xl0), "v"(xr0), "v"(xl1), "v"(xr1)); return; }
        // range diagnostic: one v_max3 (|a|, |b|, m) and one unordered compare per two values
        auto note = [&](float a, float b) {
            xmaxf_ = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)), xmaxf_);
            xnan_ |= __builtin_isunordered(a, b);
        };
        if (fuse_in) {   // padding (rows / edge voxels outside the volume) must stay 0: its (scale, shift) pair is zeroed, and
                         // max(0*x + 0, floor) = 0 for both floors — no select, no branch
            const int gd = s.d0 - 1 + sdl, gh = s.h0 - 1 + shl;
            const float mid = ((unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H) ? x_scale : 0.0f;
            const float rf = relu_floor * x_scale;
            const float *const tb = aff + (s.chunk * 16 + team * 8 + 2 * cp) * 2;   // (scale, shift) of channel 2cp, then of 2cp+1
            if constexpr (part < 2) {
                const float2 sc = *reinterpret_cast<const float2 *>(tb + 2 * part);
                const float m = sc.x * mid, a = sc.y * mid;
                f32x4 &xa = part == 0 ? xa0 : xa1, &xb = part == 0 ? xb0 : xb1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    xa[i] = fmaxf(xa[i] * m + a, rf);
                    xb[i] = fmaxf(xb[i] * m + a, rf);
                }
            } else {
                const float lft = s.w0 > 0 ? mid : 0.0f, rgt = s.w0 + PP_TW < W ? mid : 0.0f;
                const float4 sc = *reinterpret_cast<const float4 *>(tb);
                xl0 = fmaxf(xl0 * (sc.x * lft) + sc.y * lft, rf);
                xr0 = fmaxf(xr0 * (sc.x * rgt) + sc.y * rgt, rf);
                xl1 = fmaxf(xl1 * (sc.z * lft) + sc.w * lft, rf);
                xr1 = fmaxf(xr1 * (sc.z * rgt) + sc.w * rgt, rf);
            }
        } else if constexpr (part == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { xa0[i] *= x_scale; xb0[i] *= x_scale; }
        } else if constexpr (part == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { xa1[i] *= x_scale; xb1[i] *= x_scale; }
        } else {
            xl0 *= x_scale; xr0 *= x_scale; xl1 *= x_scale; xr1 *= x_scale;
        }
        if constexpr (part == 0) { note(xa0[0], xa0[1]); note(xa0[2], xa0[3]); note(xb0[0], xb0[1]); note(xb0[2], xb0[3]); }
        else if constexpr (part == 1) { note(xa1[0], xa1[1]); note(xa1[2], xa1[3]); note(xb1[0], xb1[1]); note(xb1[2], xb1[3]); }
        else { note(xl0, xr0); note(xl1, xr1); }
    };
    // output pair q of the row: F(2,3) input transform (fp32, after the scale), hi/lo split, 8 half2 stores into the team's buffer.
    // Split of t: hi = rne_f16(t), lo = rne_f16(t - hi) with t - hi as ONE v_fma_mix_f32 (f16 source, exact); the four positions are
    // written as four independent chains (stage by stage, not position by position): a LOAD phase has ~250 cycles for this slice, and the
    // dependent form (packed adds, f16 -> f32 conversions) took 430.
    unsigned char *const xw = smem + PP_LDS_X + team * PP_XBUF_B + srow * PP_XROW_B + cp * 4;
    auto halo_write = [&](auto Qc) __attribute__((always_inline)) {
        constexpr int q = decltype(Qc)::value;
        if (PP_ABL & 1) return;
        auto v0 = [&](int i) -> float { return i == 0 ? xl0 : i <= 4 ? xa0[i - 1] : i <= 8 ? xb0[i - 5] : xr0; };   // (i folds: q is a constant)
        auto v1 = [&](int i) -> float { return i == 0 ? xl1 : i <= 4 ? xa1[i - 1] : i <= 8 ? xb1[i - 5] : xr1; };
        const float t0[4] = {v0(2 * q) - v0(2 * q + 2), v0(2 * q + 1) + v0(2 * q + 2), v0(2 * q + 2) - v0(2 * q + 1), v0(2 * q + 1) - v0(2 * q + 3)};
        const float t1[4] = {v1(2 * q) - v1(2 * q + 2), v1(2 * q + 1) + v1(2 * q + 2), v1(2 * q + 2) - v1(2 * q + 1), v1(2 * q + 1) - v1(2 * q + 3)};
        unsigned hv[4], lv[4];
        float l0[4], l1[4];
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) hv[pp] = pp_cvt_pk(t0[pp], t1[pp]);
        if constexpr (!SINGLE) {
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                l0[pp] = pp_sub_lo(hv[pp], t0[pp]);
                l1[pp] = pp_sub_hi(hv[pp], t1[pp]);
            }
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) lv[pp] = pp_cvt_pk(l0[pp], l1[pp]);
        }
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
#if PP_ABL & 128   /* timing only: the arithmetic without the LDS stores */
            asm volatile("" ::"v"(hv[pp]));
            if constexpr (!SINGLE) asm volatile("" ::"v"(lv[pp]));
#else
            *reinterpret_cast<unsigned *>(xw + pp * PP_XPOS_B + q * PP_XPAIR_B) = hv[pp];
            if constexpr (!SINGLE) *reinterpret_cast<unsigned *>(xw + PP_XPART_B + pp * PP_XPOS_B + q * PP_XPAIR_B) = lv[pp];
#endif
        }
    };
    auto load_aff = [&](int n, int first, int stride) {
        for (int i = first; i < Ci * 2; i += stride) aff[i] = in_affine[(size_t)n * Ci * 2 + i];
    };
    // the same table for another frame, as LDS-DMA (ONE wave, up to three pieces of 1 KiB; lanes beyond the table are masked off): nothing in
    // the main loop is a compiler-visible load, so hipcc never waits vmcnt(0) in it
    const unsigned aff_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem + PP_LDS_AFF;
    auto dma_aff = [&](int n) __attribute__((always_inline)) {
        const unsigned char *const src = reinterpret_cast<const unsigned char *>(in_affine + (size_t)n * Ci * 2);
        const int bytes = Ci * 8;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i * 1024 < bytes && lane * 16 + i * 1024 < bytes)
                asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"((unsigned)(lane * 16 + i * 1024)), "s"(src), "s"(aff_lds + i * 1024) : "memory");
    };

    // ---- weight stream ------------------------------------------------------------------------------------------------------------
    // piece i of a wave covers LDS bytes [(wave + 8 i) KiB, +1 KiB) of the slab image; a lane's 16 bytes lie in k-group block
    // (part, position, kg) = o / 1536 at channel (o % 1536) / 16.  Its source: the same (part, position, channel) of k-group block
    // item / 9 of the pack's slab (chunk16, item % 9), item = 2 * step + kg.
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    // Lanes of k-group block kg = 1 read one slab (24576 B) further than those of kg = 0 — item 2s+1 is the next tap of the same k-group
    // of the pack — except at step 4, where item 8 is (tap 8, k-group 0) and item 9 (tap 0, k-group 1): two per-lane offset sets, no
    // per-step address arithmetic (the step's tap goes into the scalar base).
    unsigned dsrc[3], dsrc4[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const unsigned o = (unsigned)(wave + 8 * i) * 1024u + (unsigned)lane * 16u;
        const unsigned blk = o / PP_KGBLK_B;
        const bool kg1 = (blk & 1u) != 0;
        const unsigned in_slab = (blk >> 1) * (2u * PP_KGBLK_B) + (o - blk * PP_KGBLK_B);
        dsrc[i] = in_slab + (kg1 ? (unsigned)PP_SLAB_B : 0u);
        dsrc4[i] = in_slab + (kg1 ? (unsigned)PP_KGBLK_B : 8u * PP_SLAB_B);
    }
    const unsigned char *const wbytes = reinterpret_cast<const unsigned char *>(wslabs) + (size_t)cot * nchunks * 9 * PP_SLAB_B;
    auto wchunk = [&](int chunk) -> const unsigned char * { return wbytes + (size_t)chunk * 9 * PP_SLAB_B; };
    auto dma_slab = [&](auto SQc, const unsigned char *base) __attribute__((always_inline)) {   // slab SQ (0..8) of the period at `base`
        constexpr int sq = decltype(SQc)::value;
        constexpr unsigned V0 = sq == 4 ? 0u : ((2 * sq) % 9) * PP_SLAB_B + ((2 * sq) / 9) * PP_KGBLK_B;   // (tap, k-group) of item 2 sq
        constexpr unsigned slot = sq % PP_R;
        if (PP_ABL & 4) return;
        const unsigned dst = lds0 + slot * PP_SLAB_B + (unsigned)wave * 1024u;
        if constexpr (sq == 4) pp_dma3(base + V0, dsrc4[0], dsrc4[1], dsrc4[2], dst, dst + 8192u, dst + 16384u);
        else pp_dma3(base + V0, dsrc[0], dsrc[1], dsrc[2], dst, dst + 8192u, dst + 16384u);
    };

    // fragment bases (bytes)
    const unsigned a_off = (unsigned)(((p * 2 + kgl) * PP_COT + j) * 16);
    // X: the two k-groups of a step read items 2s and 2s+1 — one halo row apart (+16 B), eight rows apart (tap (kd,2) -> (kd+1,0): +128 B), or
    // (step 4) the last tap of buffer 0 and the first of buffer 1: three per-lane bases, the step's own offset is an immediate
    const unsigned b_lane = (unsigned)(PP_LDS_X + p * PP_XPOS_B + (j & 3) * PP_XPAIR_B + ((2 * team) * PP_HH + (j >> 2)) * PP_XROW_B);
    const unsigned b_row = b_lane + (unsigned)kgl * PP_XROW_B, b_plane = b_lane + (unsigned)kgl * (8u * PP_XROW_B),
                   b_buf = b_lane + (unsigned)kgl * (unsigned)(PP_XBUF_B - pp_rowoff(8) * PP_XROW_B);

    PPROF_DECL;
    // ---- prologue: slabs 0 and 1 in flight, both halves of period 0 staged ------------------------------------------------------------
    PpPeriod cur = period_at(j_first, c_begin), nxt = cur;
    bool nxt_ok = per_total > 1;
    if (nxt_ok) nxt = period_next(cur);
    if (tid < PP_COT) reinterpret_cast<float *>(smem + PP_LDS_BIAS)[tid] = (gridDim.z == 1 && bias) ? bias[cot * PP_COT + tid] : 0.0f;
    if (fuse_in) {
        load_aff(cur.n, tid, 512);
        aff_n = cur.n;
        lds_barrier();
    }
    dma_slab(std::integral_constant<int, 0>{}, wchunk(cur.chunk));
    dma_slab(std::integral_constant<int, 1>{}, wchunk(cur.chunk));
    halo_load(cur, std::integral_constant<int, 0>{});
    halo_load(cur, std::integral_constant<int, 1>{});
    halo_load(cur, std::integral_constant<int, 2>{});
    lds_dma_wait<0>();
    __builtin_amdgcn_sched_barrier(0);
    halo_convert(cur, std::integral_constant<int, 0>{});
    halo_convert(cur, std::integral_constant<int, 1>{});
    halo_convert(cur, std::integral_constant<int, 2>{});
    halo_write(std::integral_constant<int, 0>{});
    halo_write(std::integral_constant<int, 1>{});
    halo_write(std::integral_constant<int, 2>{});
    halo_write(std::integral_constant<int, 3>{});
    lds_dma_wait<0>();
    lds_barrier();

    half8 ah[3], al[3], bh[2], bl[2];
    const float unscale = whdr[0] * x_unscale;
    const int co0 = cot * PP_COT;
    const bool direct = gridDim.z == 1;
    float *const Ex = reinterpret_cast<float *>(smem + (team == 0 ? PP_LDS_EX : PP_LDS_X + PP_XBUF_B));   // [position][slot 0..5][lane][4]
    int gp = 0;   // period being multiplied
    int epi_stores = 0;   // global stores of the last epilogue: younger than the weight pieces the next two LOAD phases wait for
    bool pre2 = false;    // slab 2 of the tile that starts was issued at the top of the previous tile's epilogue (see there)
#if PP_PRIO == 2
    if (team == 1) __builtin_amdgcn_s_setprio(1);
#endif
    if (team == 1) lds_barrier();   // the stagger: team B runs one phase behind team A

    for (int tj = j_first; tj < ntiles; tj += (int)gridDim.x) {
        const int en = cur.n, ed0 = cur.d0, eh0 = cur.h0, ew0 = cur.w0, etile = tile_at(tj);
        int tz = 0;
        asm volatile("" : "+v"(tz));  // opaque 0, new per tile: keeps the epilogue's per-channel address math / bias loads out of the tile loop
        f32x16 acc[3][2];
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.0f;

        for (int c = c_begin; c < c_end; ++c, ++gp) {
            // here: cur = the period being multiplied, nxt = its successor (if nxt_ok)
            const unsigned char *const wcur = wchunk(cur.chunk);
            const unsigned char *const wnxt = wchunk(nxt.chunk);
            const bool reload_aff = fuse_in && nxt_ok && nxt.n != aff_n;   // (uniform)
            auto step = [&](auto SPc) __attribute__((always_inline)) {
                constexpr int sp = decltype(SPc)::value;
                // ---------------- LOAD(sp): everything but the MFMAs ----------------
                PPROF_STAMP(2 * sp)
#ifdef MPHIP_PP_PROFILE
                if (sp == 0 && pet_) { pe_[6] += pt1_ - pet_; pet_ = 0; }
#endif
                // An LDS-DMA piece needs ~1 us to land under load (MI355X_MICROARCH.md "ldsdma-fill"): this step's pieces (slab sp+2) are issued
                // early in the phase (right behind its fragment reads) and waited for at the END of the wave's next LOAD phase — two full phases later; that phase's barrier
                // publishes them one phase before their first reader.
                const bool issue = !(sp == 0 && pre2) && gp * 9 + sp + 2 < s_total;
                auto dma = [&]() __attribute__((always_inline)) {
                    if (issue) {
                        if constexpr (sp + 2 < 9) dma_slab(std::integral_constant<int, sp + 2>{}, wcur);
                        else dma_slab(std::integral_constant<int, sp + 2 - 9>{}, wnxt);
                    }
                };
#if PP_PRIO == 3
                __builtin_amdgcn_s_setprio(1);
#endif
                if (!(PP_ABL & 32) || gp == 0) {
                    constexpr int I0 = 2 * sp, I1 = 2 * sp + 1;
                    constexpr unsigned off0 = (I0 / 9) * PP_XBUF_B + pp_rowoff(I0 % 9) * PP_XROW_B;
                    constexpr unsigned off1 = (I1 / 9) * PP_XBUF_B + pp_rowoff(I1 % 9) * PP_XROW_B;
                    const unsigned char *const wsl = smem + (sp % PP_R) * PP_SLAB_B + a_off;
#pragma unroll
                    for (int m = 0; m < 3; ++m) {
                        ah[m] = *reinterpret_cast<const half8 *>(wsl + m * 512);
                        if constexpr (!SINGLE) al[m] = *reinterpret_cast<const half8 *>(wsl + PP_WPART_B + m * 512);
                    }
                    static_assert(off1 - off0 == PP_XROW_B || off1 - off0 == 8 * PP_XROW_B || off1 - off0 == PP_XBUF_B - pp_rowoff(8) * PP_XROW_B, "k-group distance");
                    const unsigned char *const xb = smem + (off1 - off0 == PP_XROW_B ? b_row : off1 - off0 == 8 * PP_XROW_B ? b_plane : b_buf) + off0;
                    bh[0] = *reinterpret_cast<const half8 *>(xb);
                    bh[1] = *reinterpret_cast<const half8 *>(xb + PP_HH * PP_XROW_B);
                    if constexpr (!SINGLE) {
                        bl[0] = *reinterpret_cast<const half8 *>(xb + PP_XPART_B);
                        bl[1] = *reinterpret_cast<const half8 *>(xb + PP_XPART_B + PP_HH * PP_XROW_B);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                // (r06: the fragment reads — what the next MFMA phase waits for — go out first, the LDS-DMA pieces behind them: the four
                //  waves of a team reach this point together, and twelve 16-byte-per-lane VMEM instructions hold the address unit ~190 cycles.
                //  Same box, three interleaved rounds: 0.457 / 0.461 / 0.462 ms with the pieces first, 0.456 / 0.455 / 0.458 ms so.)
                dma();
                __builtin_amdgcn_sched_barrier(0);
                // Halo staging, one slice per LOAD phase.  In the team's own count ts (A: ts = step; B starts at step 4: ts = step - 4 mod 9):
                //   ts 0, 1, 2: load part 0 (channel 2cp), part 1 (channel 2cp+1), part 2 (the four edge voxels)
                //   ts 3, 4   : normalise / scale parts 0, 1 (their loads were retired by the counted waits that ended ts 2 and ts 3)
                //   ts 5      : part 2, then output pair 0;   ts 6, 7, 8: output pairs 1, 2, 3
                // A stages the next period's even half into buffer 0 (free during steps 5-8); B the odd half into buffer 1 (free during steps
                // 0-3 of the period the data belongs to: B's ts 5-8 fall into the NEXT period, where that data is `cur`).
                // hl_[1 + ts]: halo loads of step ts; HA / HB: those of this and the previous LOAD phase — all younger than the pieces waited for
                constexpr int hl_[10] = {0, 2, 2, 4, 0, 0, 0, 0, 0, 0};
                constexpr int spa = sp, spb = (sp + 5) % 9;
                constexpr int HA = hl_[1 + spa] + hl_[spa], HB = hl_[1 + spb] + hl_[spb];
                constexpr int ts = team == 0 ? spa : spb;
                constexpr bool late = team == 1 && sp < 4;                  // (B's ts 5-8)
                const PpPeriod &stp = late ? cur : nxt;
                const bool st_on = late ? gp > 0 : nxt_ok;
                const bool hl_on = nxt_ok && !(PP_ABL & 2);
                const bool dma_on = issue && !(PP_ABL & 4);
                if (st_on) {
                    if constexpr (ts < 3) halo_load(stp, std::integral_constant<int, (ts < 3 ? ts : 0)>{});
                    else if constexpr (ts < 5) halo_convert(stp, std::integral_constant<int, (ts >= 3 && ts < 5 ? ts - 3 : 0)>{});
                    else {
                        if constexpr (ts == 5) halo_convert(stp, std::integral_constant<int, 2>{});
                        halo_write(std::integral_constant<int, (ts >= 5 ? ts - 5 : 0)>{});
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                // The wait that ends a LOAD phase: this wave's pieces of slab sp+1 (issued first thing in its previous LOAD phase) have landed
                // — its barrier below publishes them.  Younger, and left in flight: this phase's three pieces, the halo loads of this and
                // the previous phase (HA / HB: per team, compile-time), and — in a tile's first phase — the epilogue's stores.
                constexpr int HT = team == 0 ? HA : HB;
                if (sp < 2 && epi_stores) {   // (uniform; vmcnt retires in order: a tile's first two waits must not ask for its predecessor's stores)
                    const int h = hl_on ? HT : 0;
                    auto wait_e = [&](auto Ec) __attribute__((always_inline)) {
                        constexpr int E = decltype(Ec)::value;
                        if (h == 0) lds_dma_wait<3 + E>(); else if (h == 2) lds_dma_wait<3 + 2 + E>(); else lds_dma_wait<3 + 4 + E>();
                    };
                    if (epi_stores == 12) wait_e(std::integral_constant<int, 12>{}); else wait_e(std::integral_constant<int, 24>{});
                    if (sp == 1) epi_stores = 0;
                } else if (!dma_on) lds_dma_wait<0>();   // (the stream's last two steps: nothing was issued, drain)
                else if (HT != 0 && hl_on) lds_dma_wait<3 + HT>();
                else lds_dma_wait<3>();
                if (sp == 0) pre2 = false;
                if constexpr (sp == 0 && team == 1) {
                    // the fused-GroupNorm table of the next period's frame: its last reader was this phase's part-2 conversion, the next one
                    // is team A's step 3, three barriers on (this wave's next counted wait retires the transfer)
                    if (reload_aff && p == 0) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        dma_aff(nxt.n);
                    }
                }
#if PP_PRIO == 3
                __builtin_amdgcn_s_setprio(0);
#endif
                __builtin_amdgcn_sched_barrier(0);
                lds_barrier();
                PPROF_STAMP(2 * sp + 1)
                // ---------------- MFMA(sp): P1 = Wlo*Xhi, P2 = Whi*Xhi, P3 = Whi*Xlo ----------------
                __builtin_amdgcn_sched_barrier(0);
#if PP_PRIO == 1
                __builtin_amdgcn_s_setprio(1);
#endif
#if PP_ABL & 16
                asm volatile("" ::"v"(ah[0]), "v"(ah[1]), "v"(ah[2]), "v"(al[0]), "v"(al[1]), "v"(al[2]), "v"(bl[0]), "v"(bl[1]), "v"(bh[0]), "v"(bh[1]));
#else
                if constexpr (!SINGLE && !(PP_ABL & 1024)) {   /* (1024: timing only — 12 of the 18 MFMAs, what a 2-D F(2x2,3x3) transform would leave, at NO extra cost) */
#pragma unroll
                    for (int m = 0; m < 3; ++m)
#pragma unroll
                        for (int t = 0; t < 2; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh[t], acc[m][t], 0, 0, 0);
                }
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh[t], acc[m][t], 0, 0, 0);
                if constexpr (!SINGLE) {
#pragma unroll
                    for (int m = 0; m < 3; ++m)
#pragma unroll
                        for (int t = 0; t < 2; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl[t], acc[m][t], 0, 0, 0);
                }
#endif
#if PP_PRIO == 1
                __builtin_amdgcn_s_setprio(0);
#endif
                __builtin_amdgcn_sched_barrier(0);
                lds_barrier();
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

        // ---- output transform + epilogue: three rounds (one 32-channel row tile each) through the team's exchange region -------------
        // (the four positions of a pair live in the four waves of a team; the teams are one barrier apart, each in its own region)
#if PP_ABL & 8
        if (tiles_total > 0) {
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int t = 0; t < 2; ++t) asm volatile("" ::"v"(acc[m][t]));
            continue;
        }
#endif
        PPROF_EPI(0)
        // The next tile's slab 2 goes out HERE, ahead of this tile's output stores (its slot — slab 8's — was last read a barrier ago): vmcnt
        // retires in order, so issued from the next LOAD phase it would sit behind the twelve stores, and the wait for it, two phases
        // later, would be a wait for a chip-wide store burst.  This way the stores have until the wait for slab 3.
        if (!(PP_ABL & 4) && gp * 9 + 2 < s_total) {
            dma_slab(std::integral_constant<int, 2>{}, wchunk(cur.chunk));
            pre2 = true;
        }
        const int gn_rows = tiles_total * 2;   // channel-major [Co][tile * 2 + plane pair][2] (the finalize kernel reads rows of it)
        if (gn_part && etile == 0 && tid == 0) gn_part[(size_t)gn_rows * Co * 2] = unscale;   // (behind the partials)
        // The rounds are instantiated per Winograd position (the wave's `p` is a run-time value, the registers it parks / keeps are not: with
        // p a variable every unit went through select chains, ~90 VALU instructions per round).  Stores: uniform base (scalar registers) +
        // one 32-bit per-lane offset for all twelve, so that no 64-bit address is assembled per store.
        const bool odd = (lane & 1) != 0;
        // (this lane's QUAD of voxels: lanes 2k / 2k+1 store the 4 voxels 4k..4k+3 of a row, for different channels)
        float *const ybase = (direct ? y : y + (size_t)blockIdx.z * N * Co * DHW) + ((size_t)en * Co + co0) * DHW + (size_t)(ed0 + 2 * team) * HW + (size_t)eh0 * W + ew0;
        const unsigned yoff = (unsigned)(((8 * p + 4 * kgl + (odd ? 2 : 0)) * DHW + (j >> 2) * W + 2 * (j & 2)) * 4);
        auto rounds = [&](auto Pc) __attribute__((always_inline)) {
            constexpr int P = decltype(Pc)::value;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                // park the units other waves finish: unit u = accumulator registers 4u..4u+3 of both column tiles; wave P keeps unit P
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (u != P) {
                            const int slot = t * 3 + (u - (u > P ? 1 : 0));
                            const f32x4 v = {acc[m][t][4 * u], acc[m][t][4 * u + 1], acc[m][t][4 * u + 2], acc[m][t][4 * u + 3]};
                            *reinterpret_cast<f32x4 *>(Ex + ((P * 6 + slot) * 64 + lane) * 4) = v;
                        }
                lds_barrier();
                PPROF_EPI(1 + 2 * m)
                float ssum[4] = {0.0f, 0.0f, 0.0f, 0.0f}, qsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                const f32x4 bv = *reinterpret_cast<const f32x4 *>(smem + PP_LDS_BIAS + (m * 32 + 8 * P + 4 * kgl) * 4);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x4 M[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (q != P) {
                            const int slot = t * 3 + (P - (P > q ? 1 : 0));
                            M[q] = *reinterpret_cast<const f32x4 *>(Ex + ((q * 6 + slot) * 64 + lane) * 4);
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
                        if (gn_part) {   // (uniform)
                            ssum[i] += r0 + r1;
                            qsum[i] = __builtin_fmaf(r0, r0, qsum[i]);
                            qsum[i] = __builtin_fmaf(r1, r1, qsum[i]);
                        }
                        y0[i] = r0 * unscale + bv[i];
                        y1[i] = r1 * unscale + bv[i];
                    }
                    // 16-byte stores: a lane holds one output pair (2 voxels) of 4 channels; lanes 2k / 2k+1 hold neighbouring pairs of a row and
                    // trade halves (quad_perm [1,0,3,2]): the even lane ends up with 4 consecutive voxels of channels 0-1, the odd lane with
                    // those of channels 2-3
#define PP_SWAP(v_) __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v_), 0xB1, 0xf, 0xf, false))
                    const float g0 = PP_SWAP(odd ? y0[0] : y0[2]), g1 = PP_SWAP(odd ? y1[0] : y1[2]);
                    const float g2 = PP_SWAP(odd ? y0[1] : y0[3]), g3 = PP_SWAP(odd ? y1[1] : y1[3]);
#undef PP_SWAP
                    const f32x4 va = {odd ? g0 : y0[0], odd ? g1 : y1[0], odd ? y0[2] : g0, odd ? y1[2] : g1};
                    const f32x4 vb = {odd ? g2 : y0[1], odd ? g3 : y1[1], odd ? y0[3] : g2, odd ? y1[3] : g3};
                    unsigned char *const dq = reinterpret_cast<unsigned char *>(ybase + (size_t)(m * 32 + tz) * DHW + (size_t)t * HW);   // (uniform)
#if PP_ABL & 512   /* timing only: the output transform without its global stores */
                    asm volatile("" ::"v"(va), "v"(vb), "s"(dq));
#else
                    *reinterpret_cast<f32x4 *>(dq + yoff) = va;
                    *reinterpret_cast<f32x4 *>(dq + (size_t)DHW * 4 + yoff) = vb;
#endif
                }
                if (gn_part) {
                    // per-channel (sum, sum of squares) of the RAW transformed accumulators over this wave's 2 x 64 voxels of the channel:
                    // the 32 lanes of a half-wave hold one channel's columns (the finalize kernel applies unscale and the bias in double)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#define PP_ROW_ADD(v_, ctrl_) v_ += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v_), ctrl_, 0xf, 0xf, false));
                        PP_ROW_ADD(ssum[i], 0x128) PP_ROW_ADD(qsum[i], 0x128)   // row_ror:8, :4, :2, :1 -> every lane of a 16-lane row: the row's sum
                        PP_ROW_ADD(ssum[i], 0x124) PP_ROW_ADD(qsum[i], 0x124)
                        PP_ROW_ADD(ssum[i], 0x122) PP_ROW_ADD(qsum[i], 0x122)
                        PP_ROW_ADD(ssum[i], 0x121) PP_ROW_ADD(qsum[i], 0x121)
#undef PP_ROW_ADD
                        // rows 1 and 3 add the totals of rows 0 and 2 (row_bcast:15, row mask 0b1010): lanes 16-31 / 48-63 hold a half-wave's sum
                        ssum[i] += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(ssum[i]), 0x142, 0xa, 0xf, false));
                        qsum[i] += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(qsum[i]), 0x142, 0xa, 0xf, false));
                    }
                    if (j == 31) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int co = co0 + m * 32 + 8 * P + 4 * kgl + i + tz;
                            *reinterpret_cast<float2 *>(gn_part + ((size_t)co * gn_rows + (size_t)etile * 2 + team) * 2) = make_float2(ssum[i], qsum[i]);
                        }
                    }
                }
                lds_barrier();   // the region is rewritten by the next round / (team B's) by the next period's halo
                PPROF_EPI(2 + 2 * m)
            }
        };
        switch (p) {   // (wave-uniform)
        case 0: rounds(std::integral_constant<int, 0>{}); break;
        case 1: rounds(std::integral_constant<int, 1>{}); break;
        case 2: rounds(std::integral_constant<int, 2>{}); break;
        default: rounds(std::integral_constant<int, 3>{}); break;
        }
        epi_stores = (PP_ABL & 512) ? (gn_part ? 12 : 0) : gn_part ? 24 : 12;
    }
    if (team == 0) lds_barrier();   // (team B's extra barrier at the start)

    // operands outside the f16 range (non-finite inputs, or finite ones beyond a wrong caller-supplied descriptor) are not clamped — they
    // propagate as Inf / NaN — but they are counted, per thread that saw any
    const bool sat_ = xnan_ || xmaxf_ > 0.5f * F16_CLAMP;
    if (__builtin_amdgcn_ballot_w64(sat_) != 0) {  // never taken in normal operation
        unsigned tot = sat_;
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) tot += __shfl_xor(tot, sft, 64);
        if (lane == 0) atomicAdd(&g_f16x3_wino_pp_saturated, (unsigned long long)tot);
    }
    PPROF_FLUSH
    };
    if (team_rt == 0) run(std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 1>{});
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
int f16x3_wino_pp_saturation(unsigned long long *count, int reset) {
    if (hipMemcpyFromSymbol(count, HIP_SYMBOL(g_f16x3_wino_pp_saturated), sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        const unsigned long long z = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_f16x3_wino_pp_saturated), &z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}

void f16x3_wino_pp_launch(dim3 grid, hipStream_t s, hipEvent_t t0, hipEvent_t t1, const float *x, const _Float16 *slabs, const float *hdr,
                          const float *bias, float *dst, int N, int Ci, int Co, int D, int H, int W, int cps, unsigned xb,
                          const float *in_affine, int in_relu, const float *x_range, int tiles, int xcd_on, const int *tile_list,
                          float *gn_part, bool half_products) {
#define PP_LAUNCH(S_)                                                                                                                    \
    {                                                                                                                                    \
        if (t0 && t1)                                                                                                                    \
            hipExtLaunchKernelGGL(conv3d_k3_f16x3_wino_pp_kernel<S_>, grid, dim3(512), 0, s, t0, t1, 0, x, slabs, hdr, bias, dst, N, Ci, Co, D, \
                                  H, W, cps, xb, in_affine, in_relu, x_range, tiles, xcd_on, tile_list, gn_part);                        \
        else                                                                                                                             \
            hipLaunchKernelGGL(conv3d_k3_f16x3_wino_pp_kernel<S_>, grid, dim3(512), 0, s, x, slabs, hdr, bias, dst, N, Ci, Co, D, H, W, cps,    \
                               xb, in_affine, in_relu, x_range, tiles, xcd_on, tile_list, gn_part);                                      \
    }
    if (half_products) PP_LAUNCH(true) else PP_LAUNCH(false)
#undef PP_LAUNCH
}

}  // namespace mphip

#ifdef MPHIP_PP_PROFILE
extern "C" int mphip_debug_wino_pp_profile(unsigned long long *out64, int reset) {
    if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(mphip::g_pp_prof), 512) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[64] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(mphip::g_pp_prof), z, 512) != hipSuccess) return -1;
    }
    return 0;
}
#endif
