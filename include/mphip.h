/*
 * mphip.h — C ABI of libmphip.so: the MI355X (gfx950) kernels behind the MegaPortraits
 * Gbase hot slice (reference: johndpope/MegaPortrait-hack model.py:1151-1171).
 *
 * The reference has no native code and no FFI: its hot path is Python calling PyTorch ATen
 * ops.  Every entry point below therefore replaces an ATen call site (or a short run of
 * them) of model.py, cited per function; the reference-side binding a maintainer would add
 * is the ctypes stub shown in INTEGRATION.md (and implemented for real in
 * megaportrait-hack_amd/_lib.py).
 *
 * Conventions
 *  - Plain C: pointers are DEVICE pointers (HBM) unless named host_*; sizes are ints.
 *    No torch types.  `stream` is a hipStream_t passed as void* (NULL = default stream).
 *  - Tensors are float32, contiguous, NCDHW (N,C,D,H,W), exactly the layout the reference's
 *    modules exchange (model.py:271 defines the volume layout: channel c*16+d -> (c,d)).
 *  - Every function returns 0 on success or a negative MPHIP_E* code, never throws, never
 *    allocates device memory, never synchronises the stream (the one diagnostic that does,
 *    mphip_f16x3_saturation_count, says so).  Work is stream-ordered;
 *    borrowed pointers must stay valid until the stream reaches the end of the call's work.
 *    mphip_last_error() returns a thread-local message for the last failing call.
 *  - Scratch memory is caller-supplied: query with the matching *_workspace_bytes().
 */
#ifndef MPHIP_H
#define MPHIP_H

#include <stddef.h>
#include <stdint.h>

#define MPHIP_RANGE_FLOATS 4100 /* floats per range descriptor: 4 + up to 4096 per-workgroup partial maxima */

#ifdef __cplusplus
extern "C" {
#endif

#define MPHIP_OK 0
#define MPHIP_EINVAL (-1)   /* bad argument (shape, null pointer, unsupported size) */
#define MPHIP_ELAUNCH (-2)  /* hipLaunchKernel / runtime error, see mphip_last_error() */
#define MPHIP_EWORKSPACE (-3) /* workspace too small */

/* ABI version of this header.  Bumped whenever an exported signature or a packed layout changes; mphip_version() returns the
 * value the LIBRARY was built with — compare the two after dlopen (the ctypes binding does, and refuses a mismatch). */
#define MPHIP_ABI_VERSION 14
int mphip_version(void);
/* hipGraph hygiene (ABI 13).  On ROCm 7.x a MEMSET node of a captured hipGraph is not reliably ordered with its neighbouring kernel nodes
 * (observed twice: stale f16x3 pack headers, r03; a training step's loss that kept its previous value, r04-r05 — ATen's multi-block
 * reduction zeroes its semaphores with hipMemsetAsync).  Rewrites `graph` (a hipGraph_t) in place: every 1-D MEMSET node becomes a kernel
 * node with the same bytes, predecessors and successors; *replaced (may be NULL) receives the count.  Call between capture and
 * instantiation.  Not stream-ordered; no effect on graphs without memset nodes.                                                        */
int mphip_graph_memsets_to_kernels(void *graph, int *replaced);
/* (ABI 14) MEMSET nodes `graph` still holds, child graphs included — the ones the rewrite above leaves alone (2-D memsets, element sizes
 * other than 1/2/4 bytes, nodes inside child graphs).  Non-zero after a rewrite = the hazard is still there: the caller should warn.   */
int mphip_graph_memset_nodes_left(void *graph, int *left);

/* Build flags of the loaded library.  Bit 0: a DEVELOPMENT variant — at least one kernel was compiled with a timing-only ablation
 * (csrc/mphip_ablate.h) and computes wrong results by design; the product build returns 0 and the Python loader refuses anything else
 * unless MPHIP_ALLOW_ABLATED=1.                                                                                                     */
int mphip_build_flags(void);
const char *mphip_last_error(void);

/* ------------------------------------------------------------------ K0  rigid transform
 * Replaces compute_rotation_matrix + the 4x4 assembly + torch.inverse of compute_rt_warp,
 * model.py:790-801 and 811-856: rotation [B,3] Euler degrees (x,y,z), translation [B,3] ->
 * theta [B,3,4] = first three rows of A=[R|t;0 0 0 1] (R = Rx@(Ry@Rz)), inverted when
 * invert != 0 (WarpGeneratorS2C, model.py:965).  On the device so the forward has no host sync. */
int mphip_rt_theta(const float *rotation_deg, const float *translation, float *theta, int B, int invert,
                   void *stream);

/* ------------------------------------------------------------------ K1  warp field compose
 * Replaces model.py:965-973 / 1016-1022 (inside WarpGeneratorS2C/C2D.forward):
 *   F.affine_grid(theta,(B,1,G,G,G),align_corners=False).permute(0,4,1,2,3)       [:804-806]
 * + F.interpolate(em,(G,G,G),'trilinear',align_corners=False)                      [:971]
 * theta [B,3,4] (already inverted for S2C), em [B,3,eD,eH,eW], base_tbl[G] =
 * torch.linspace(-1,1,G)*(G-1)/G built on the host -> w [B,3,G,G,G].
 * If rt_out / em_out are non-NULL the two addends are also written (same shape) for tests. */
int mphip_warp_field_compose(const float *theta, const float *em, const float *base_tbl, float *w,
                             float *rt_out, float *em_out, int B, int eD, int eH, int eW, int G,
                             void *stream);

/* ------------------------------------------------------------------ K2  volumetric warp
 * Replaces apply_warping_field(v, warp_field), model.py:1028-1065:
 *   F.interpolate(field,(D,H,W),'trilinear',align_corners=True) + linspace identity grid
 *   + 2*g/(S-1)-1 + F.grid_sample(v, grid, 'bilinear', 'border', align_corners=True).
 * v [B,C,D,H,W], field [B,3,fD,fH,fW], lin_d/h/w = torch.linspace(-1,1,S) host-built tables
 * on the device -> out [B,C,D,H,W].  Two launches: a coordinate pass (the bit-exact index
 * chain, 12 B per voxel into `workspace`, mphip_warp_workspace_bytes()) and the gather pass
 * (source bounding box of each tile staged in LDS).  Optional outputs for the index contract:
 * coords_out [B,D,H,W,3] float (clipped x,y,z; when given it replaces the workspace),
 * idx_out [B,D,H,W,3] int32 (floor indices; requires coords_out). */
size_t mphip_warp_workspace_bytes(int B, int D, int H, int W);
/* Optional, K2 only (mphip_warp_volume / mphip_warp_volume_coords): a workspace that is this many bytes LARGER than the entry point's
 * minimum lets the gather bring the volume's low corner — voxels [0,6)^3, where every sample of the reference's own fields lies
 *  — into LDS from one compact copy per frame (one contiguous LDS-DMA transfer per workgroup, issued under its
 * coordinate loads) instead of collecting it from 96 channel planes itself.  Results are bit-identical either way. */
size_t mphip_warp_corner_image_bytes(int B, int C);
/* ... or built on its own, early (the hot slice has `v` long before the coordinates), and handed to the gather (img: 16-byte aligned,
 * mphip_warp_corner_image_bytes(B, C) bytes, layout private to the library; workspace as mphip_warp_volume_coords) */
int mphip_warp_corner_image(const float *v, void *img, size_t img_bytes, int B, int C, int D, int H, int W, void *stream);
int mphip_warp_volume_coords_img(const float *v, const float *coords, float *out, float *out_range, int B, int C, int D, int H, int W,
                                 void *workspace, size_t workspace_bytes, const void *img, void *stream);
/* out_range (optional, MPHIP_RANGE_FLOATS floats): range descriptor of `out` (see "Range descriptors" below) — the gather pass folds
 * max|out| in, so the conv that consumes the warped volume (G3d's first, model.py:1160) needs no extra pass. */
int mphip_warp_volume(const float *v, const float *field, const float *lin_d, const float *lin_h,
                      const float *lin_w, float *out, float *coords_out, int32_t *idx_out, float *out_range, int B,
                      int C, int D, int H, int W, int fD, int fH, int fW, void *workspace, size_t workspace_bytes,
                      void *stream);

/* ------------------------------------------------------------------ K3  warp + depth projection
 * Replaces apply_warping_field (model.py:1167) fused with torch.sum(dim=2) (model.py:1171):
 * out [B,C,H,W] = sum_d warp(v, field)[b,c,d,h,w]; the warped volume is never written.     */
int mphip_warp_volume_dsum(const float *v, const float *field, const float *lin_d, const float *lin_h,
                           const float *lin_w, float *out, int B, int C, int D, int H, int W, int fD,
                           int fH, int fW, void *workspace, size_t workspace_bytes, void *stream);
/* Cross-reenactment (BASELINE config 5, the reference recomputes G3d per pair: inference.py:35 / model.py:1160-1171):
 * ONE source volume v [1,C,D,H,W] warped by B driver fields [B,3,fD,fH,fW] -> out [B,C,H,W]; the volume is read in
 * place (no B-fold expanded copy).  Same workspace as mphip_warp_volume_dsum.                          */
int mphip_warp_volume_dsum_shared(const float *v, const float *field, const float *lin_d, const float *lin_h,
                                  const float *lin_w, float *out, int B, int C, int D, int H, int W, int fD,
                                  int fH, int fW, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------ K4/K5  Conv3d k=3 / k=1
 * Replaces nn.Conv3d(Ci,Co,3,padding=1) (model.py:505,507,591,374-375,458) and
 * nn.Conv3d(Ci,Co,1) / nn.Conv2d 1x1 (model.py:510,380,446) — stride 1, bias.
 * x [N,Ci,D,H,W] fp32 -> y [N,Co,D,H,W] fp32 in both precisions.
 * precision 0: exact fp32 (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain), any shape.
 * precision 1: "f16x3" — operands split into two f16 halves, 3 f16 MFMAs per product, fp32
 *              accumulate (~2^-21 relative per term, fp32 class, for tensors of ANY magnitude: every
 *              operand tensor is scaled by its own power of two, see "Range descriptors").  Only for
 *              k=3, Ci%16==0, Co%96==0, H%8==0, W%8==0, D%2==0: ask mphip_conv3d_supported().
 *
 * Range descriptors (precision 1).  x_range = MPHIP_RANGE_FLOATS floats on the device describing the magnitude of x:
 *   [0] scale  [1] 1/scale  [2] max|x| (or a rigorous upper bound)  [3] n (as uint32)  [4 .. 4+n) partial maxima
 *   [0] != 0: explicit power-of-two operand scale (mphip_grad_prep writes gradients' this way; 4 floats suffice then);
 *   [0] == 0: the kernel derives it from m = max([2], partials) — the power of two with m*scale in [2^13, 2^14).
 * Producers that already stream the tensor fill one for free — one partial maximum per workgroup, plain stores, no
 * atomics (mphip_warp_volume, mphip_groupnorm_apply*, mphip_groupnorm_affine_table: `out_range` arguments);
 * mphip_absmax_range() makes one for a tensor of unknown origin, and x_range == NULL lets the conv do that itself (one
 * extra read of x; MPHIP_RANGE_FLOATS*4 bytes of workspace).  A caller who knows a bound B can hand in {0, 0, B, 0}.
 * With a correct descriptor no finite value can leave the f16 range — the reference's fp32 conv has no range cliff,
 * neither has this; Inf/NaN inputs (and finite ones beyond a WRONG descriptor) are not clamped, they propagate as
 * Inf/NaN and are counted (mphip_f16x3_saturation_count).  precision 0 ignores x_range.
 * Weights are used in a packed layout built once per weight version and per precision:
 *   precision 0: OIDHW -> [k^3][CiP][CoP] fp32 (CoP = Co up to 32, CiP = Ci up to 2, zero padded)
 *   precision 1: 16-byte header (1/scale, scale) + the LDS image of every (96-channel tile,
 *                16-channel chunk, 3-tap group) slab as f16 hi/lo planes.
 * workspace: MPHIP_RANGE_FLOATS*4 bytes for a library-computed range descriptor (precision 1) + split-K partial
 * sums for small volumes; mphip_conv3d_workspace_bytes() gives the total.                     */
int mphip_conv3d_supported(int N, int Ci, int Co, int D, int H, int W, int k, int precision);
size_t mphip_packed_weight_bytes(int Co, int Ci, int k, int precision);
int mphip_pack_conv_weight(const float *w_oidhw, void *w_packed, int Co, int Ci, int k, int precision,
                           void *stream);
size_t mphip_conv3d_workspace_bytes(int N, int Ci, int Co, int D, int H, int W, int k, int precision);
int mphip_absmax_range(const float *x, size_t n, float *range, void *stream);   /* x 16-byte aligned */
int mphip_conv3d_fwd(const float *x, const float *x_range, const void *w_packed, const float *bias, float *y, int N,
                     int Ci, int Co, int D, int H, int W, int k, int precision, void *workspace,
                     size_t workspace_bytes, void *stream);

/* conv + the statistics of the GroupNorm that follows it (nn.Conv3d -> nn.GroupNorm pairs at
 * model.py:505-508, 390-399, 458-460): same y as mphip_conv3d_fwd plus gn_stats [N*gn_groups][2] =
 * (mean, rstd) of y, identical to mphip_groupnorm_stats(y) — one host call for the conv -> GN pair. */
size_t mphip_conv3d_gn_workspace_bytes(int N, int Ci, int Co, int D, int H, int W, int k, int precision,
                                       int gn_groups);
int mphip_conv3d_gn_fwd(const float *x, const float *x_range, const void *w_packed, const float *bias, float *y, float *gn_stats,
                        int N, int Ci, int Co, int D, int H, int W, int k, int precision, int gn_groups,
                        float gn_eps, void *workspace, size_t workspace_bytes, void *stream);

/* GroupNorm folded into the NEXT conv (nn.GroupNorm -> ReLU -> nn.Conv3d, model.py:506-507/517/518 and
 * 390-396): mphip_groupnorm_affine_table turns (mean, rstd) + gamma/beta (+ AdaptiveGroupNorm's w2/b2) into
 * table[N][C][2] = (scale, shift); mphip_conv3d_gnin_fwd applies x' = x*scale + shift (then ReLU if in_relu)
 * to every in-volume voxel while it stages its input tile (padding stays 0), so the normalised tensor is never
 * written.  precision 1 (f16x3) only, Ci <= 768.  The table call also fills out_range, the range descriptor of the
 * NORMALISED tensor the conv will see, from a data-independent bound: |x-mean|*rstd <= sqrt(elements per group), so
 * |x'[c]| <= sqrt(C/G*S)*|gamma*w2| + |beta*w2 + b2| (S = D*H*W of x); pass it as x_range.               */
int mphip_groupnorm_affine_table(const float *stats, const float *gamma, const float *beta, const float *w2,
                                 const float *b2, float *table, float *out_range, int N, int C, int S, int G,
                                 void *stream);
/* conv + statistics + the affine table / range bound of the norm that follows, in one call (saves the table launch per block) */
int mphip_conv3d_gn_table_fwd(const float *x, const float *x_range, const void *w_packed, const float *bias, float *y, float *gn_stats,
                              const float *gamma, const float *beta, const float *w2, const float *b2, float *table, float *table_range,
                              int N, int Ci, int Co, int D, int H, int W, int k, int precision, int gn_groups, float gn_eps,
                              void *workspace, size_t workspace_bytes, void *stream);
/* FlowField's first three residual blocks (reference model.py:369-408 at the fixed levels of model.py:439-471) in two launches per block:
 *   y = upsample_nearest(relu?(AGN(conv3x3x3(x) + b) [+ conv1x1x1(res_x) + res_b]))      AGN: GroupNorm(groups) * gamma + beta, then * w2 + b2
 * One workgroup owns a whole (frame, GroupNorm group): full input-channel loop, statistics, both affines, residual conv, ReLU and the
 * nearest upsample in ONE launch, from the ORIGINAL weights (w [Co,Ci,3,3,3], res_w [Co,Cr,1,1,1]; no packed copy).  Shapes: the levels
 * of FlowField only — (Co, D, H, W) in {(256,4,1,1), (128,8,2,2), (64,16,4,4)}, 32 groups, Ci (and Cr) a multiple of the level's
 * slice count (128, 128, 16); _supported returns the level (1-3) or 0.  x [N,Ci,D,H,W], res_x [N,Cr,D,H,W] or NULL (Cr = 0),
 * y [N,Co,D*uD,H*uH,W*uW]. */
int mphip_flowfield_conv_gn_supported(int Ci, int Co, int D, int H, int W, int Cr, int groups);
int mphip_flowfield_conv_gn(const float *x, const float *w, const float *b, const float *gamma, const float *beta, const float *w2,
                            const float *b2, const float *res_x, const float *res_w, const float *res_b, float *y, int N, int Ci, int Co,
                            int D, int H, int W, int Cr, int uD, int uH, int uW, int groups, float eps, int relu, void *stream);
/* Level 1 (4x1x1 volumes: only the taps (kd, 1, 1) exist) from a compact copy [Co][Ci][3] of the weights — same bits, a ninth of the weight traffic;
 * mphip_flowfield_compact_weight_bytes() is 0 for every other level. */
size_t mphip_flowfield_compact_weight_bytes(int Ci, int Co, int D, int H, int W);
int mphip_flowfield_compact_weight(const float *w, void *w_compact, int Ci, int Co, void *stream);
int mphip_flowfield_conv_gn_compact(const float *x, const void *w_compact, const float *b, const float *gamma, const float *beta, const float *w2,
                                    const float *b2, const float *res_x, const float *res_w, const float *res_b, float *y, int N, int Ci, int Co,
                                    int D, int H, int W, int Cr, int uD, int uH, int uW, int groups, float eps, int relu, void *stream);
/* FlowField's output head (model.py:458-465): em = tanh(relu(GroupNorm(1, 3)(Conv3d(32, 3, 3)(x)))), x [N,32,16,16,16] -> em [N,3,16,16,16]:
 * a direct 3-channel conv (one workgroup per depth slice, partial sums in double) + one normalising pass; w [3,32,3,3,3] as stored. */
size_t mphip_flowfield_out_workspace_bytes(int N);
int mphip_flowfield_out(const float *x, const float *w, const float *b, const float *gamma, const float *beta, float *em, int N, float eps,
                        void *workspace, size_t workspace_bytes, void *stream);
/* measurement: the next conv launch on this thread carries these two HIP events (hipEvent_t).  The f16x3 3x3x3 kernels are launched
 * with them attached (hipExtLaunchKernelGGL: they take the kernel's own begin / end — time spent waiting for CUs that another stream's
 * kernel holds is not counted); every other conv kernel has them recorded on the launch stream right before / after it. */
void mphip_conv3d_time_next_launch(void *event_begin, void *event_end);
int mphip_conv3d_gnin_fwd(const float *x, const float *in_affine, const float *x_range, int in_relu, const void *w_packed,
                          const float *bias, float *y, int N, int Ci, int Co, int D, int H, int W, int k,
                          int precision, void *workspace, size_t workspace_bytes, void *stream);
int mphip_conv3d_gnin_gn_fwd(const float *x, const float *in_affine, const float *x_range, int in_relu, const void *w_packed,
                             const float *bias, float *y, float *gn_stats, int N, int Ci, int Co, int D, int H, int W,
                             int k, int precision, int gn_groups, float gn_eps, void *workspace,
                             size_t workspace_bytes, void *stream);

/* Split-K aware chain for small volumes (G3d's 4x16x16 / 2x8x8 levels, all of FlowField): the conv leaves
 * its result as `splits` partial slabs out[z][N,Co,D,H,W] (bias NOT added) and the GroupNorm statistics /
 * apply kernels that consume it sum the slabs on the fly (z ascending + bias, the same order as the
 * reduce of mphip_conv3d_fwd), so the separate reduce launch and pass disappear.
 * mphip_conv3d_splits() == 1 means the conv is not split: out is then the plain result (bias added).
 * workspace: only a library-computed range descriptor (precision 1 and x_range == NULL).                  */
int mphip_conv3d_splits(int N, int Ci, int Co, int D, int H, int W, int k, int precision);
int mphip_conv3d_fwd_split(const float *x, const float *x_range, const void *w_packed, const float *bias, float *out,
                           int N, int Ci, int Co, int D, int H, int W, int k, int precision, void *workspace,
                           size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------ K6  GroupNorm
 * Replaces nn.GroupNorm(G,C) eps=1e-5 (model.py:506,508,460,309) and what the reference
 * applies right after it.  Two steps so the global per-(sample,group) reduction is explicit:
 *  stats: x [N,C,S] -> stats [N*G][2] = (mean, rstd), biased variance.
 *         workspace: mphip_groupnorm_workspace_bytes(N,C,S,G).
 *  apply: y = (x-mean)*rstd*gamma[c]+beta[c];  if w2: y = y*w2[c]+b2[c]  (AdaptiveGroupNorm,
 *         model.py:314-316);  if residual: y += residual  (model.py:522);  if relu: max(y,0)
 *         (model.py:517,523);  if tanh_: y = tanh(y) after relu (model.py:462-465).
 *         pool2 != 0 additionally averages 2x2x2 cells (nn.AvgPool3d(2,2), model.py:576-580):
 *         then y is [N,C,D/2,H/2,W/2] and D,H,W must be given (S = D*H*W).
 *         out_range (optional): range descriptor of y for the conv that reads it (max|y| folded in). */
size_t mphip_groupnorm_workspace_bytes(int N, int C, int S, int G);
int mphip_groupnorm_stats(const float *x, float *stats, int N, int C, int S, int G, float eps,
                          void *workspace, size_t workspace_bytes, void *stream);
int mphip_groupnorm_apply(const float *x, const float *stats, const float *gamma, const float *beta,
                          const float *w2, const float *b2, const float *residual, float *y, float *out_range, int N,
                          int C, int D, int H, int W, int G, int relu, int tanh_, int pool2, void *stream);

/* Split-aware variants (see mphip_conv3d_fwd_split): x and/or residual may be `*_splits` partial slabs with
 * their conv bias passed separately; the output can additionally be nearest-upsampled by (uD,uH,uW) —
 * the nn.Upsample that follows each FlowField block (model.py:427-433, 450-457) — or 2x2x2 pooled.
 * Meant for small tensors (one thread per output element; stats: group span <= 65536 floats).        */
int mphip_groupnorm_stats_split(const float *x, int x_splits, const float *x_bias, float *stats, int N, int C,
                                int S, int G, float eps, void *stream);
int mphip_groupnorm_apply_split(const float *x, int x_splits, const float *x_bias, const float *stats,
                                const float *gamma, const float *beta, const float *w2, const float *b2,
                                const float *residual, int res_splits, const float *res_bias, float *y,
                                float *out_range, int N, int C, int D, int H, int W, int G, int relu, int tanh_,
                                int pool2, int uD, int uH, int uW, void *stream);

/* statistics + apply in ONE launch for tiny tensors (a (sample, group) span of <= 12288 floats: every FlowField
 * layer): one workgroup per (sample, group), values cached in LDS between the two passes.  Same arguments
 * as the split-aware pair; stats_out (optional) receives (mean, rstd).                                   */
int mphip_groupnorm_small_fused(const float *x, int x_splits, const float *x_bias, const float *gamma,
                                const float *beta, const float *w2, const float *b2, const float *residual,
                                int res_splits, const float *res_bias, float *y, float *stats_out, int N, int C,
                                int D, int H, int W, int G, float eps, int relu, int tanh_, int uD, int uH, int uW,
                                void *stream);

/* ------------------------------------------------------------------ K7  resampling
 * avgpool2:            nn.AvgPool3d(2,2)                                   (model.py:576-580)
 * upsample_trilinear2: nn.Upsample(scale_factor=2,'trilinear',align_corners=True) (:585-589)
 * upsample_nearest:    nn.Upsample(scale_factor=(sD,sH,sW)) default nearest (:427-433)   */
int mphip_avgpool2(const float *x, float *y, int NC, int D, int H, int W, void *stream);
int mphip_upsample_trilinear2(const float *x, float *y, int NC, int D, int H, int W, void *stream);
int mphip_upsample_nearest(const float *x, float *y, int NC, int D, int H, int W, int sD, int sH, int sW,
                           void *stream);
/* F.interpolate(x, scale_factor=(sD,sH,sW), 'trilinear', align_corners=False), integer factors: the `upsample=True` branch of
 * ResBlock3D / ResBlock3D_Adaptive (model.py:404-405, 525-526), and its adjoint (dx [NC,D,H,W], 16-byte aligned; fp32 atomics). */
int mphip_upsample_trilinear(const float *x, float *y, int NC, int D, int H, int W, int sD, int sH, int sW, void *stream);
int mphip_upsample_trilinear_bwd(const float *dout, float *dx, int NC, int D, int H, int W, int sD, int sH, int sW,
                                 void *stream);

/* ------------------------------------------------------------------ K8  small head ops
 * mphip_add_matmul: out[b,n] = sum_k (a[b,k]+a2[b,k]) * m[k,n]   — (z+e) @ Gamma, model.py:945-957
 *                   (a2 may be NULL; trans!=0 uses m[n,k] and adds bias[n]: the 1x1 Conv2d on a
 *                   1x1 map at model.py:446).                                              */
int mphip_add_matmul(const float *a, const float *a2, const float *m, const float *bias, float *out, int B,
                     int K, int N, int trans, void *stream);

/* ------------------------------------------------------------------ K9  backward (training, scope row f2)
 * Gradients of the G3d building blocks as torch.autograd computes them for the reference's modules.
 * grad_prep:         one pass over a conv's output gradient dy [N,C,S]: dbias[c] = sum dy (may be NULL) and
 *                    scale[0..2] = (s, 1/s, max|dy|), s = the power of two with max|dy|*s in [2^13, 2^14) — the f16x3 kernels'
 *                    operand scale for this tensor (scale: 4 floats of device memory).
 * conv3d_bwd_data:   dx = conv(dy, Wt), Wt[ci][co][a][b][c] = W[co][ci][k-1-a][k-1-b][k-1-c]; same kernels as
 *                    mphip_conv3d_fwd (Ci/Co = channels of dy/dx).  mphip_pack_conv_weight_bwd_data packs Wt straight
 *                    from the original OIDHW weight W (its Co/Ci = Wt's: the original conv's Ci/Co); `_like` reuses the scale header
 *                    of W's own f16x3 forward pack instead of scanning W again.
 * conv3d_bwd_weight: dW[co][ci][tap] = sum_{n,v} dY[n][co][v] * X[n][ci][v+tap]  (nn.Conv3d model.py:505-510, 591).
 *                    precision 0: exact fp32 MFMA, any shape, k in {1,3}; precision 1: f16x3 (k=3: W%8==0; k=1: D*H*W%128==0).
 *                    Split over voxels, deterministic slab reduce.
 * groupnorm_bwd_reduce / _apply: nn.GroupNorm (+ residual + ReLU) backward (model.py:506-523).  reduce computes
 *                    s1[n][c] = sum du, s2[n][c] = sum du*xhat (du = dy*(y>0) when relu) and folds them into
 *                    dbeta[c] = sum_n s1, dgamma[c] = sum_n s2, ab[n][g] = (sum_c gamma*s1, sum_c gamma*s2)/count;
 *                    apply writes dx = rstd*(gamma*du - a - xhat*b) and dres = du (dres may be NULL).
 *                    y = the forward output (activation mask), stats = the forward (mean, rstd).
 *                    act: 0 none, 1 ReLU, 2 tanh(ReLU(.)) (model.py:462-465).  w2 (may be NULL) = AdaptiveGroupNorm's
 *                    second affine (model.py:314-316): gamma_eff = gamma*w2, and dw2/db2 are produced.
 * upsample_nearest_bwd: adjoint of mphip_upsample_nearest (D,H,W = dims of dx).
 * small_gemm:        out[m][n] = sum_k (a[m*sam+k*sak] (+a2)) * b[k*sbk+n*sbn] (+bias[n]) — the generators' dense heads
 *                    ((z+e)@Gamma, the 1x1 conv on a 1x1 map) and their gradients; strides in elements.
 * avgpool2_bwd, upsample_trilinear2_bwd: adjoints of K7 (D,H,W = dims of dx).                      */
size_t mphip_grad_prep_workspace_bytes(int N, int C, int S);
int mphip_grad_prep(const float *dy, float *dbias, float *scale, int N, int C, int S, void *workspace,
                    size_t workspace_bytes, void *stream);
int mphip_pack_conv_weight_bwd_data(const float *w, void *wp, int Co, int Ci, int k, int precision, void *stream);
int mphip_pack_conv_weight_bwd_data_like(const float *w, void *wp, int Co, int Ci, int k, int precision,
                                         const void *fwd_pack, void *stream);
int mphip_conv3d_bwd_data(const float *dy, const void *wt_packed, float *dx, const float *dy_scale, int N, int Ci,
                          int Co, int D, int H, int W, int k, int precision, void *workspace, size_t workspace_bytes,
                          void *stream);
/* Batched re-packing (ABI 13): every conv weight of a module in at most five launches instead of two or three per weight — a training
 * step re-packs all of them after its optimizer update (train.py:283-284 -> the next forward), and ~100 dependent 5-20 us launches are a
 * millisecond of a 10 ms step.  A job is one mphip_pack_conv_weight / _bwd_data / _bwd_data_like call: `transposed` selects the bwd-data
 * direction (Co / Ci are then the bwd-data conv's own, as in mphip_pack_conv_weight_bwd_data), `like` (precision 1 only, may be NULL) is
 * a pack of the SAME weight tensor — another job's `wp` or an existing pack — whose header (max|w|) is reused instead of a second
 * reduction.  The table resolves the jobs once (pointers and shapes are fixed; the weights' VALUES are read at every run), owns a small
 * device buffer (hipMalloc / hipFree in create / destroy: not stream-ordered, not capturable) and mphip_pack_table_run only launches
 * kernels on `stream` (capturable).  Results are bit-identical to the single calls.                                               */
typedef struct mphip_pack_job {
    const float *w;      /* OIDHW weight (device)                                                             */
    void *wp;            /* destination pack, mphip_packed_weight_bytes(Co, Ci, k, precision) bytes (device)  */
    const void *like;    /* see above                                                                         */
    int Co, Ci, k, precision, transposed;
} mphip_pack_job;
int mphip_pack_table_create(const mphip_pack_job *jobs, int n, void **table);
int mphip_pack_table_run(void *table, void *stream);
int mphip_pack_table_destroy(void *table);
int mphip_conv3d_bwd_weight_supported(int N, int Ci, int Co, int D, int H, int W, int k, int precision);
size_t mphip_conv3d_bwd_weight_workspace_bytes(int N, int Ci, int Co, int D, int H, int W, int k, int precision);
/* x_range: the range descriptor the forward conv used for x (precision 1; NULL = computed here, one extra read of x) */
int mphip_conv3d_bwd_weight(const float *x, const float *x_range, const float *dy, const float *dy_scale, float *dw, int N,
                            int Ci, int Co, int D, int H, int W, int k, int precision, void *workspace,
                            size_t workspace_bytes, void *stream);
size_t mphip_groupnorm_bwd_workspace_bytes(int N, int C, int S);
int mphip_groupnorm_bwd_reduce(const float *x, const float *y, const float *dy, const float *stats,
                               const float *gamma, const float *beta, const float *w2, float *dgamma, float *dbeta,
                               float *dw2, float *db2, float *ab, int N, int C, int S, int G, int act,
                               void *workspace, size_t workspace_bytes, void *stream);
int mphip_groupnorm_bwd_apply(const float *x, const float *y, const float *dy, const float *stats,
                              const float *gamma, const float *w2, const float *ab, float *dx, float *dres, int N,
                              int C, int S, int G, int act, void *stream);
/* reduce + apply in two launches instead of three (ABI 13): the fold of the partial sums is re-derived by every apply workgroup for its
 * own (frame, group) — same operation order, same bits as mphip_groupnorm_bwd_reduce + mphip_groupnorm_bwd_apply.  dres, w2 / beta / dw2 /
 * db2 may be NULL (no residual branch; no second affine).                                                                          */
int mphip_groupnorm_bwd(const float *x, const float *y, const float *dy, const float *stats, const float *gamma, const float *beta,
                        const float *w2, float *dx, float *dres, float *dgamma, float *dbeta, float *dw2, float *db2, int N, int C, int S,
                        int G, int act, void *workspace, size_t workspace_bytes, void *stream);
int mphip_avgpool2_bwd(const float *dout, float *dx, int NC, int D, int H, int W, void *stream);
size_t mphip_upsample_trilinear2_bwd_workspace_bytes(int NC, int D, int H, int W);
/* workspace: three separable bandwidth passes (D, H, W); NULL: one gather pass that needs no scratch (slower on large tensors) */
int mphip_upsample_trilinear2_bwd(const float *dout, float *dx, int NC, int D, int H, int W, void *workspace,
                                  size_t workspace_bytes, void *stream);
int mphip_upsample_nearest_bwd(const float *dout, float *dx, int NC, int D, int H, int W, int sD, int sH, int sW,
                               void *stream);
int mphip_small_gemm(const float *a, const float *a2, const float *b, const float *bias, float *out, int M, int N,
                     int K, long sam, long sak, long sbk, long sbn, void *stream);

/* ------------------------------------------------------------------ K10  backward of K0-K3 (training, row f2)
 * warp_coords:            the coordinate pass of K2/K3 alone: coords[B,D,H,W,3] (clipped x,y,z sample positions).
 * warp_volume_bwd:        backward of mphip_warp_volume (dsum=0) / mphip_warp_volume_dsum (dsum=1; dout is [B,C,H,W]):
 *                         dv [B,C,D,H,W] and dfield [B,3,fD,fH,fW] (ATen rule: clipped coordinates pass no gradient;
 *                         then the adjoint of the align_corners=True resize, model.py:1036).  Either may be NULL.
 *                         Per frame: if every sample of the frame falls into a box of <= 5^3 source voxels (the
 *                         reference's own fields) dv is a deterministic fp32-MFMA reduction over the outputs and the
 *                         coordinate gradient comes from an LDS image of the box; otherwise dv is a scatter with
 *                         hardware fp32 atomics (like the reference's grid_sample backward on a GPU).
 * warp_field_compose_bwd: dtheta [B,3,4] (F.affine_grid backward) and dem [B,3,eD,eH,eW] (adjoint of the
 *                         align_corners=False resize, model.py:971-973) from dw [B,3,G,G,G].  Either may be NULL.
 * rt_theta_bwd:           (drot [B,3] in degrees, dtr [B,3]) from dtheta through the rotation composition and,
 *                         if invert, the matrix inverse (model.py:790-801, 811-856).                        */
int mphip_warp_coords(const float *field, const float *lin_d, const float *lin_h, const float *lin_w, float *coords,
                      int B, int D, int H, int W, int fD, int fH, int fW, void *stream);
size_t mphip_warp_volume_bwd_workspace_bytes(int B, int C, int D, int H, int W);
int mphip_warp_volume_bwd(const float *v, const float *field, const float *lin_d, const float *lin_h,
                          const float *lin_w, const float *dout, float *dv, float *dfield, int B, int C, int D, int H,
                          int W, int fD, int fH, int fW, int dsum, void *workspace, size_t workspace_bytes,
                          void *stream);
size_t mphip_warp_field_compose_bwd_workspace_bytes(int B, int G);
int mphip_warp_field_compose_bwd(const float *dw, const float *base_tbl, float *dtheta, float *dem, int B, int eD,
                                 int eH, int eW, int G, void *workspace, size_t workspace_bytes, void *stream);
int mphip_rt_theta_bwd(const float *rot, const float *tr, const float *dtheta, float *drot, float *dtr, int B,
                       int invert, void *stream);

/* K1 + K2/K3's coordinate pass in one launch, and the gathers on given coordinates: the composed warp field [B,3,G,G,G] is read
 * exactly once in the reference's flow (by the warp it feeds, model.py:965-973 -> 1154, 1016-1022 -> 1167), which needs only 2*D of
 * its G depth planes — mphip_warp_field_coords evaluates the same expressions at those planes: bit-identical coordinates
 * [B,D,G,G,3], no field in HBM.  mphip_warp_volume_coords = K2's gather passes on them (workspace: mphip_warp_workspace_bytes). */
int mphip_warp_field_coords(const float *theta, const float *em, const float *base_tbl, const float *lin_d, const float *lin_h,
                            const float *lin_w, float *coords, int B, int eD, int eH, int eW, int G, int D, void *stream);
int mphip_warp_volume_coords(const float *v, const float *coords, float *out, float *out_range, int B, int C, int D, int H, int W,
                             void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------ demand-driven evaluation of a gather's producer
 * G3d's last conv feeds apply_warping_field + torch.sum(dim=2) (model.py:1160 -> 1167-1171): a gather whose sample positions
 * depend only on the C2D warp field, known long before the conv runs.  Only the voxels the gather reads need to exist:
 *   mphip_warp_coords(field, ...) -> coords [B,D,H,W,3]                  (K3's coordinate pass, the bit-exact index chain)
 *   mphip_warp_sample_box(coords) -> box [B][8] = {lx, ly, lz, ex, ey, ez, -, -}: per frame the box of source voxels the samples
 *                                     touch (all 8 trilinear corners, zero-weight ones included)
 *   mphip_upsample_trilinear2_roi / mphip_conv3d_fwd_roi: produce only the output tiles (of the conv kernel's own tiling,
 *                                     mphip_conv3d_roi_granule) a box touches, and of the upsample only those tiles' halos; every
 *                                     other voxel of y is UNSPECIFIED (left untouched by an unsplit launch; a split-K shape's ordered reduce
 *                                     writes every voxel, unlisted ones from uninitialised slabs).  roi_frames == 0: box b belongs to
 *                                     frame b; roi_frames > 0: N == 1 volume serves that many boxes (1 source x many drivers).
 *                                     Shapes / precisions without a tiled kernel (granule() == 0) compute everything.
 *   mphip_warp_volume_dsum_coords(v, coords) = mphip_warp_volume_dsum with the coordinate pass already done.
 * The values that are computed are bit-identical to the full evaluation; nothing else is ever read.  On the reference's
 * fields (samples inside the ~5^3 low corner) the box covers 2 of the 256 tiles of a 16x64x64 frame; a field
 * that travels through the volume makes the box the volume, i.e. the full conv.  The hot-slice plan uses this for final_conv. */
int mphip_warp_sample_box(const float *coords, int *box, int B, int D, int H, int W, void *stream);
int mphip_conv3d_roi_granule(int N, int Ci, int Co, int D, int H, int W, int k, int precision, int *tile_dhw);
size_t mphip_conv3d_roi_workspace_bytes(int N, int Ci, int Co, int D, int H, int W, int k, int precision);   /* conv workspace + the tile list */
int mphip_conv3d_fwd_roi(const float *x, const float *x_range, const void *w_packed, const float *bias, float *y, const int *roi,
                         int roi_frames, int N, int Ci, int Co, int D, int H, int W, int k, int precision, void *workspace,
                         size_t workspace_bytes, void *stream);
int mphip_upsample_trilinear2_roi(const float *x, float *y, const int *roi, int roi_frames, int N, int C, int D, int H, int W, int tD,
                                  int tH, int tW, void *stream);
int mphip_warp_volume_dsum_coords(const float *v, const float *coords, float *out, int B, int C, int D, int H, int W, int shared,
                                  void *stream);
/* Training: the gather's gradient wrt its input volume is exactly zero outside the same boxes (mphip_warp_volume_bwd), so the
 * backward of the producing conv can be restricted too: bwd-data zero-fills dx and computes only the tiles the boxes grown by one
 * voxel touch (workspace: mphip_conv3d_roi_workspace_bytes with Ci/Co of dy/dx); bwd-weight skips the voxel tiles outside the boxes. */
int mphip_conv3d_bwd_data_roi(const float *dy, const void *wt_packed, float *dx, const float *dy_scale, const int *roi, int N, int Ci,
                              int Co, int D, int H, int W, int k, int precision, void *workspace, size_t workspace_bytes, void *stream);
int mphip_conv3d_bwd_weight_roi(const float *x, const float *x_range, const float *dy, const float *dy_scale, float *dw,
                                const int *dy_boxes, int N, int Ci, int Co, int D, int H, int W, int k, int precision, void *workspace,
                                size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------ one-call entries: the hot slice and G3d as plans
 * Replace the single calls of the reference: Gbase.forward's slice (model.py:1151-1171: WarpGeneratorS2C ->
 * apply_warping_field -> G3d -> WarpGeneratorC2D -> apply_warping_field -> torch.sum(dim=2)) and G3d.forward
 * (model.py:593-597).  A plan is built from the reference's state-dict: `names[i]` = the key relative to Gbase
 * ("warp_generator_s2c.flowfield.resblock1.conv1.weight", "G3d.downsampling.0.gn1.bias", "G3d.final_conv.weight", ...; the
 * keys of WarpGeneratorS2C / WarpGeneratorC2D / G3d listed in SURVEY.md Appendix C, `adaptive_matrix_beta` is not read —
 * the reference never uses it, model.py:958-963), `tensors[i]` = that tensor on the device, fp32, contiguous, in the
 * reference's own layout (OIDHW conv weights).  The plan keeps the POINTERS (the caller keeps the tensors alive) and
 * owns what it derives from them: packed conv weights (hipMalloc, built on the first forward's stream), the side stream
 * on which the C2D generator runs underneath G3d, and the carve-up of the caller's workspace.  The launches are the
 * per-op entry points above in the order the Python host issues them: results are bitwise identical to that path.
 *   flags: MPHIP_PLAN_G3D_ONLY — only the G3d.* keys are needed (mphip_g3d_forward only);
 *          MPHIP_PLAN_SINGLE_STREAM — no side stream (everything on the caller's stream).
 *   C,D,H,W: the appearance volume (96,16,64,64 in the reference, model.py:1157).  torch.linspace(-1,1,n) tables for
 *          n = 16 and 64 and F.affine_grid's 64^3 base grid are built in (bit patterns captured from the reference's CPU
 *          path); for other sizes hand the device tables in with mphip_hot_slice_plan_set_tables before the first forward.
 *   (plan sizes are cached per batch size and precision.)
 *   refresh: after an in-place weight update call it with (NULL, NULL, 0) — every pack is rebuilt on the next forward's
 *          stream; after the parameters moved (new storage) pass the new name/pointer lists.
 *   forward: vs [B,96,D,H,W], es/zs/zd [B,512], Rs/Rd [B,3] Euler degrees, ts/td [B,3] -> out [B,96,H,W]; any B (more than
 *          MPHIP_PLAN_MAX_FRAMES_PER_PASS frames run as consecutive passes).  No allocation, no host sync: capturable in a
 *          hipGraph after one warm-up call (the first call of a batch size allocates the packed weights).  Two forwards
 *          of one plan in flight on different streams need different workspaces — and share the plan's ONE side stream, so
 *          their second generator chains run one after the other: a caller that keeps several batches in flight (a serving
 *          loop; bench.py's default) creates one plan per stream (measured: +7 % with two plans, +2-3 % with one shared).
 *   mphip_g3d_forward: x [B,96,D,H,W] (+ its range descriptor or NULL) -> y [B,96,D,H,W], B <= MPHIP_PLAN_MAX_FRAMES_PER_PASS. */
typedef struct mphip_hot_slice_plan mphip_hot_slice_plan;
#define MPHIP_PLAN_G3D_ONLY 1
#define MPHIP_PLAN_SINGLE_STREAM 2
#define MPHIP_PLAN_FULL_FINAL_CONV 4   /* evaluate G3d's last upsample + conv everywhere (default: demand-driven, see above) */
#define MPHIP_PLAN_MAX_FRAMES_PER_PASS 64   /* the conv kernels address their input through one 2 GiB buffer resource */
int mphip_hot_slice_plan_create(const char *const *names, const void *const *tensors, int n_tensors, int C, int D, int H, int W,
                                int flags, mphip_hot_slice_plan **out);
int mphip_hot_slice_plan_set_tables(mphip_hot_slice_plan *plan, const float *lin_d, const float *lin_h, const float *lin_w,
                                    const float *affine_base);
/* conv arithmetic of the plan's launches: 1 (default) = "auto", f16x3 wherever mphip_conv3d_supported(..., 1), exact fp32 elsewhere;
 * 0 = exact fp32 MFMA everywhere.  Changes the workspace size: query mphip_hot_slice_workspace_bytes again. */
int mphip_hot_slice_plan_set_precision(mphip_hot_slice_plan *plan, int precision);
/* Measurement: with profiling on, every launch of the dominant conv (Conv3d 3x3x3 96->96 at the plan's full volume) is bracketed by
 * HIP events on the stream it is launched on (the conv alone: the GroupNorm statistics of its output are launched after the closing
 * event).  _read sums the durations recorded since the last read — kind 0: full launches, 1: the demand-driven launch of G3d's
 * last conv — and synchronises on them.  Not for captured forwards. */
int mphip_hot_slice_plan_profile(mphip_hot_slice_plan *plan, int enable);
int mphip_hot_slice_plan_profile_read(mphip_hot_slice_plan *plan, int kind, double *sum_ms, int *count);
int mphip_hot_slice_plan_refresh(mphip_hot_slice_plan *plan, const char *const *names, const void *const *tensors, int n_tensors);
size_t mphip_hot_slice_workspace_bytes(mphip_hot_slice_plan *plan, int B);
int mphip_hot_slice_forward(mphip_hot_slice_plan *plan, const float *vs, const float *es, const float *Rs, const float *ts,
                            const float *zs, const float *Rd, const float *td, const float *zd, float *out, int B, void *workspace,
                            size_t workspace_bytes, void *stream);
size_t mphip_g3d_workspace_bytes(mphip_hot_slice_plan *plan, int B);
int mphip_g3d_forward(mphip_hot_slice_plan *plan, const float *x, const float *x_range, float *y, int B, void *workspace,
                      size_t workspace_bytes, void *stream);
void mphip_hot_slice_plan_destroy(mphip_hot_slice_plan *plan);

/* The reference's reduced-precision policy for the convs (train.py:145,188: the generator step runs under torch.cuda.amp.autocast(), its
 * conv3d calls take f16 operands with fp32 accumulation).  mphip_conv3d_set_half_products(1) makes the CALLING THREAD's subsequent precision-1
 * 3x3x3 launches that run in the F(2,3) domain (mphip_conv3d_kernel_variant == 5: G3d's levels 0-2, Eapp's 3-D tail, their bwd-data
 * convs) and the 3x3x3 f16x3 backward-weight launches (mphip_conv3d_bwd_weight, precision 1; since ABI 13) use ONE f16 product per
 * multiply — operands rounded to f16 (in the transformed domain for the F(2,3) kernels), fp32 accumulate, a third of the matrix
 * work; every other kernel keeps its fp32-class arithmetic (autocast permits more precision, never less).  The flag is thread-local, read
 * when a launch is issued, and returns its previous value; the Python host sets it only while torch.is_autocast_enabled() with float16.
 * Results then carry f16-operand rounding (~1e-3 relative), NOT the 1e-3 max-abs fp32 contract of the default mode.                  */
int mphip_conv3d_set_half_products(int enable);

/* Diagnostic (synchronous, not stream-ordered): operand elements of the f16x3 conv kernels whose scaled value was outside
 * the f16 range since the last reset — Inf/NaN inputs, or finite values beyond a wrong caller-supplied range descriptor.
 * They are NOT clamped (the output carries Inf/NaN like the reference's fp32 conv would); 0 in normal operation.     */
int mphip_f16x3_saturation_count(unsigned long long *count, int reset);

/* Which kernel a full launch of mphip_conv3d_fwd takes for this shape: 0 = exact fp32 kernels / the k = 1 GEMM, 1 = the direct
 * f16x3 kernel on (td,8,8) tiles (2 = its (4,8,16)-tile form, removed in r05: never returned), 5 = the f16x3 kernel in the 1-D Winograd F(2,3) domain (2/3 of the MFMAs;
 * conv3d_f16x3_wino.hip).  For measurements and tests: results do not depend on it beyond fp32 rounding.                     */
int mphip_conv3d_kernel_variant(int N, int Ci, int Co, int D, int H, int W, int k, int precision);

/* Measurement only (tools/mfma_sol.py, bench.py `roofline.sustained_peak`): the f16x3 convs' MFMA stream and nothing else — three
 * v_mfma_f32_32x32x16_f16 per product on random hi/lo fragments, 3 x 2 accumulator tiles per wave, 8 waves per workgroup; mode 1 reads
 * its ten fragments per tap from LDS like the conv, mode 0 keeps them in registers.  No global traffic.  One launch issues
 * workgroups * 8 * iters * 162 MFMAs (32768 FLOP each); sink: >= workgroups * 512 floats (keeps the result alive).        */
int mphip_debug_mfma_sol(float *sink, int workgroups, int iters, int mode, void *stream);

/* Measurement only (tools/dma_stream.py): LDS-DMA streaming of an L2-resident tensor of `slabs_in_tensor` 24-KiB slabs the way the F(2,3) conv
 * streams its weights (8 waves x 3 pieces per slab, ring of four, a wave waits for the pieces it issued `lag` slabs ago, optional barrier).      */
int mphip_debug_dma_stream(const void *weights, int slabs_in_tensor, int slabs, int lag, int barrier, float *sink, int workgroups, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MPHIP_H */
