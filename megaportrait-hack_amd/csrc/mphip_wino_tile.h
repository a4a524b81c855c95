// Shared by the two F(2,3) conv kernels that use the 4x8x8-voxel tile, the 8-channel X double buffer and the [part][position][kg][co][8]
// weight slabs: conv3d_f16x3_wino_pp.hip (role-split schedule, r05) and conv3d_f16x3_wino_bt.hip (one wave per SIMD, r06).  Tile
// geometry, the LDS map, the hand-issued halo loads and the hi/lo split helpers — one definition, so that the two kernels stay
// byte-compatible (same packs, same staged X image, bit-identical results).
#pragma once
#include "mphip_conv.h"
#include "mphip_f16x3.h"

namespace mphip {

constexpr int PP_COT = 96;
constexpr int PP_R = 3;                                   // slabs in the ring
constexpr int PP_SLAB_B = 2 * 4 * 2 * PP_COT * 8 * 2;     // [part][position][kg][co][8] f16 = 24576 B
constexpr int PP_WPART_B = PP_SLAB_B / 2;
constexpr int PP_KGBLK_B = PP_COT * 16;                   // one (part, position, kg) block: 1536 B
constexpr int PP_TD = 4, PP_TH = 8, PP_TW = 8;
constexpr int PP_HH = PP_TH + 2;
constexpr int PP_ROWS = (PP_TD + 2) * PP_HH;              // 60 halo rows
constexpr int PP_XPOS_B = PP_ROWS * 4 * 16;               // (part, position) block: 240 (pair, row) slots x 8 channels = 3840 B
// Inside a (part, position) block the slots are PAIR-major: [output pair 0..3][halo row 0..59][8 channels] (r06).  A staging thread owns
// (channel pair, row) and writes 4 bytes per (position, pair): with rows 16 bytes apart the 64 lanes of a wave (4 channel pairs x 16 rows)
// cover 64 consecutive banks.  r05's row-major form ([row][pair]: rows 64 bytes apart) made every one of those stores a 4-way bank
// conflict — ALL of the kernel's 17.3 M conflict cycles per launch, 28 % of its LDS-active cycles (profiles/NOTES_r06.md).  The
// fragment reads (16 bytes per lane, lane = (row j >> 2, pair j & 3)) stay conflict-free: the 16 lanes the LDS serves together hold 4 rows
// x 4 pairs, whose first banks (48 pair + 4 row) mod 64 are 16 different multiples of 4.
constexpr int PP_XROW_B = 16;                             // halo row to halo row
constexpr int PP_XPAIR_B = PP_ROWS * PP_XROW_B;           // output pair to output pair: 960 B
constexpr int PP_XPART_B = 4 * PP_XPOS_B;
constexpr int PP_XBUF_B = 2 * PP_XPART_B;                 // one 8-channel buffer: 30720 B
constexpr int PP_EX_B = 4 * 6 * 64 * 16;                  // one team's exchange round: 24576 B
constexpr int PP_AFF_CI = 384;
constexpr int PP_LDS_X = PP_R * PP_SLAB_B;
constexpr int PP_LDS_EX = PP_LDS_X + 2 * PP_XBUF_B;
constexpr int PP_LDS_AFF = PP_LDS_EX + PP_EX_B;
constexpr int PP_LDS_BIAS = PP_LDS_AFF + PP_AFF_CI * 2 * 4;   // the workgroup's 96 bias values (read by the epilogue through LDS: a global
                                                              // load there would make hipcc wait vmcnt(0) — for the previous round's stores)
constexpr int PP_LDS_BYTES = PP_LDS_BIAS + PP_COT * 4;
static_assert(PP_XBUF_B >= PP_EX_B, "team B's exchange lives in X buffer 1");
static_assert(PP_LDS_BYTES <= 163840 - 128, "LDS");

__device__ constexpr int pp_rowoff(int tap) { return (tap / 3) * PP_HH + tap % 3; }   // halo-row offset of a (kd,kh) tap

// hi/lo split helpers as single instructions (hipcc has no builtin for either and, left alone, SLP-packs the surrounding fp32 arithmetic
// into v_pk_*_f32 — an anti-lever beside MFMAs, MI355X_MICROARCH.md — and converts hi back with two v_cvt_f32_f16):
//   pp_cvt_pk: {rne_f16(a), rne_f16(b)};  pp_sub_lo / pp_sub_hi: t - (float)h.lo / h.hi, the f16 read in place (exact: one rounding)
__device__ __forceinline__ unsigned pp_cvt_pk(float a, float b) {
    unsigned r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float pp_sub_lo(unsigned h, float t) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(t));
    return r;
}
__device__ __forceinline__ float pp_sub_hi(unsigned h, float t) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(t));
    return r;
}

// Halo loads, hidden from hipcc's waitcnt pass on purpose.  The two teams run ONE program half a period apart and share the staging
// registers: with compiler-visible loads the pass sees team A's pending loads on the path where team B transforms ITS (long landed) row and
// puts `s_waitcnt vmcnt(0)` there — team B then drains the LDS-DMA pieces it issued a moment ago, ~1 us in every write phase (and vice
// versa).  Ordering is by hand instead: the counted wait that ends each LOAD phase leaves only the newest operations in flight, so a
// unit's loads have landed two phases before its first use (cdna_hip_programming.md 5.7: no use of the destination before that wait,
// every phase ends in a sched_barrier; the kernel stays below the VGPR limit without spills, so no live range is split or copied).
typedef unsigned pp_u32x4 __attribute__((ext_vector_type(4)));
// (one statement per group of loads, opened by `s_nop 4` for the descriptor — see pp_dma3; early-clobber outputs: a destination must not
//  share a register with the address of a later load of the same statement)
__device__ __forceinline__ void pp_buf_load_2x4(pp_u32x4 rsrc, unsigned o0, unsigned o1, f32x4 &r0, f32x4 &r1) {
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %4, 0 offen\n\tbuffer_load_dwordx4 %1, %3, %4, 0 offen"
                 : "=&v"(r0), "=&v"(r1) : "v"(o0), "v"(o1), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void pp_buf_load_4x1(pp_u32x4 rsrc, unsigned o0, unsigned o1, unsigned o2, unsigned o3, float &r0, float &r1, float &r2, float &r3) {
    asm volatile("s_nop 4\n\tbuffer_load_dword %0, %4, %8, 0 offen\n\tbuffer_load_dword %1, %5, %8, 0 offen\n\t"
                 "buffer_load_dword %2, %6, %8, 0 offen\n\tbuffer_load_dword %3, %7, %8, 0 offen"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(rsrc) : "memory");
}

__device__ __forceinline__ void pp_buf_load_1x4(pp_u32x4 rsrc, unsigned o0, f32x4 &r0) {
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen" : "=&v"(r0) : "v"(o0), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void pp_buf_load_2x1(pp_u32x4 rsrc, unsigned o0, unsigned o1, float &r0, float &r1) {
    asm volatile("s_nop 4\n\tbuffer_load_dword %0, %2, %4, 0 offen\n\tbuffer_load_dword %1, %3, %4, 0 offen"
                 : "=&v"(r0), "=&v"(r1) : "v"(o0), "v"(o1), "s"(rsrc) : "memory");
}

struct PpPeriod {   // what a 16-channel period of the K stream addresses (wave-uniform)
    int n, d0, h0, w0, chunk, tj;
};

}  // namespace mphip
