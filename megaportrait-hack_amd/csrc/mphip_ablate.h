// Every compile-time switch that changes what a kernel COMPUTES (timing-only ablations: wrong results by design) or adds
// instrumentation to it, in one place.  The product build (build.sh) defines none of them; dev variants are built by
// tools/build_variant.sh / tools/k2_ablate.sh into build_variants/ and never replace libmphip.so.  (r06: the ablations of the r02-r04 kernels
// — MPHIP_ABL_* of the direct conv, MPHIP_WN_ABL_* of the lockstep F(2,3) conv — were deleted from the sources; what they measured is in
// profiles/NOTES_r01-r04.md.  The generated code of those two translation units is unchanged, instruction for instruction.)  A translation unit that is compiled
// with any ablation ON leaves a marker in the library: mphip_build_flags() reports it and the Python loader (_lib.py) REFUSES such a
// library unless MPHIP_ALLOW_ABLATED=1 is set (the dev tools set it) — a timing variant cannot become the product by a stray MPHIP_LIB.
//
//  switch                                         | file                       | effect (timing only unless noted)
//  -----------------------------------------------+----------------------------+---------------------------------------------------------
//  PP_ABL=bitmask                                 | conv3d_f16x3_wino_pp.hip   | role-split F(2,3) conv: 1 no halo convert + LDS writes, 2 no halo loads, 4 no
//                                                 |                            | weight DMA, 8 no epilogue, 16 no MFMAs, 32 no fragment reads, 128 no LDS stores,
//                                                 |                            | 512 no output stores, 1024 12 of 18 MFMAs per step (the 2-D F(2x2,3x3) bound)
//  BT_ABL=bitmask                                 | conv3d_f16x3_wino_bt.hip   | big-tile F(2,3) conv: 1 no halo convert + LDS writes, 2 no halo loads, 4 no weight
//                                                 |                            | DMA, 8 no epilogue, 16 no MFMAs, 32 no fragment reads, 64 no per-step wait + barrier
//  MPHIP_K2_ABL_NOSTAGE / _NOLOOP / _NOSTORE      | warp.hip                   | K2 without image staging / gather loop / output stores
//  -----------------------------------------------+----------------------------+---------------------------------------------------------
//  instrumentation, results unchanged (not "ablated"; slower):
//  MPHIP_PROFILE_PHASES                           | conv3d_f16x3(.hip,_wino)   | per-phase cycle counters (mphip_debug_f16x3_profile)
//  MPHIP_PP_PROFILE, PP_PRIO=n                    | conv3d_f16x3_wino_pp.hip   | per-phase wall stamps (mphip_debug_wino_pp_profile); s_setprio policy
//  MPHIP_BT_PROFILE                               | conv3d_f16x3_wino_bt.hip   | cycles per wave: whole kernel / epilogues / prologue (mphip_debug_wino_bt_profile)
//  MPHIP_WN_TRACE / MPHIP_K2_TRACE                | conv3d_f16x3_wino / warp   | per-tile / per-workgroup wall-clock traces
//  MPHIP_F16X3_OLD_FRAGS, MPHIP_BUILTIN_DMA,      | conv3d_f16x3.hip,          | r01's fragment schedule; compiler-issued LDS-DMA; halo write in its own phase
//    MPHIP_WN_NO_OVERLAP8                         | conv3d_f16x3_wino.hip      |   (same results, other schedules)
#pragma once

#ifndef PP_ABL
#define PP_ABL 0
#endif
#ifndef BT_ABL
#define BT_ABL 0
#endif

#if defined(MPHIP_K2_ABL_NOSTAGE) || defined(MPHIP_K2_ABL_NOLOOP) || defined(MPHIP_K2_ABL_NOSTORE) || (PP_ABL != 0) || (BT_ABL != 0)
#define MPHIP_ABLATED 1
// (weak: every ablated translation unit may define it; api.hip tests for its presence)
extern "C" __attribute__((weak, visibility("default"))) int mphip_ablated_build_marker = 1;
#else
#define MPHIP_ABLATED 0
#endif
