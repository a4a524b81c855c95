// Shared between the conv translation units (conv3d.hip: exact fp32 kernels + C ABI; conv3d_f16x3.hip).
#pragma once
#include "mphip_common.h"

namespace mphip {

constexpr unsigned OOB = 0x80000000u;  // >= num_records -> buffer load returns 0 (zero padding)

__device__ __forceinline__ float buf_load_f(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, (int)soff, 0));
}

typedef float mphip_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ mphip_f32x4 buf_load_f4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(mphip_f32x4, (u32x4_)__builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0));
}

struct F16x3Plan {
    int td, variant, splits, chunks_per_split;
    dim3 grid;
};

bool f16x3_supported(int N, int Ci, int Co, int D, int H, int W, int k);
size_t f16x3_packed_bytes(int Co, int Ci);
size_t f16x3_packed_bytes_k1(int Co, int Ci);
int f16x3_launch_k1(const float *x, const void *wpacked, const float *bias, float *dst, int N, int Ci, int Co, int DHW,
                    const float *x_range, hipStream_t s);
F16x3Plan f16x3_plan(int N, int Ci, int Co, int D, int H, int W, bool roi = false);
int f16x3_pack(const float *w_oidhw, void *out, int Co, int Ci, int k, int transposed, const void *header_from, hipStream_t s);
// Batched re-packing (mphip_pack_table_*): one launch per kernel kind for every weight of a module.  PackJob is the device-side job
// (the public mphip_pack_job + what the host resolved); PackSel a launch's selection: indices into the job array and the first block
// of each selected job (n + 1 entries), both device arrays.
struct PackJob {
    const float *w;
    void *wp;
    const void *like;      // precision 1: a pack of the same weight whose header (max|w|) is reused, or nullptr
    size_t wino_off;       // precision 1, k = 3: byte offset of the F(2,3) slabs in wp, 0 = none
    int Co, Ci, k, precision, transposed, reserved;
};
struct PackSel {
    const int *job;
    const int *first;
    int n, blocks;
};
int f16x3_pack_blocks(const PackJob &j, int kind /* 0 absmax, 1 k = 3 pack, 2 k = 1 pack */);
size_t f16x3_pack_wino_offset(int Co, int Ci);
int f16x3_pack_many(const PackJob *jobs, PackSel absmax, PackSel k3, PackSel k1, hipStream_t s);
// roi (optional): 8 ints per box {lx,ly,lz,ex,ey,ez,-,-}: only the output tiles a box touches are computed (roi_frames == 0: one box per
// frame; > 0: the conv's frames... single frame serves that many boxes)
int f16x3_launch(const F16x3Plan &p, const float *x, const void *wpacked, const float *bias, float *dst, int N, int Ci,
                 int Co, int D, int H, int W, const float *in_affine, int in_relu, const float *x_range, hipStream_t s,
                 const int *roi = nullptr, int roi_frames = 0, int *tile_list = nullptr /* 1 + plan.grid.x ints when roi */, int roi_dilate = 0,
                 float *gn_part = nullptr /* [Co][plan.grid.x][f16x3_tile_waves][2]: per-wave (sum, sumsq) of the output, splits == 1 only */,
                 hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr /* stamped with the conv kernel's own begin / end */);
void f16x3_tile_dims(const F16x3Plan &p, int dims[3]);
int f16x3_tile_waves(const F16x3Plan &p);

// conv3d_f16x3_wino.hip: the same split-f16 arithmetic in the 1-D Winograd F(2,3) domain (2/3 of the MFMAs).  Its slabs live
// behind the direct slabs of a precision-1 k=3 pack (0 bytes when the layer can never take the kernel); F16x3Plan.variant 4.
size_t f16x3_wino_packed_bytes(int Co, int Ci);
bool f16x3_wino_usable(int N, int Ci, int Co, int D, int H, int W);
int f16x3_wino_splits(int N, int Ci, int Co, int D, int H, int W);
long f16x3_wino_tiles(int N, int D, int H, int W);   // tiles of a launch (a depth-2 volume: one tile = two frames)
int f16x3_wino_saturation(unsigned long long *count, int reset);
int f16x3_wino_launch(const float *x, const void *slabs, const float *hdr, const float *bias, float *dst, int N, int Ci, int Co, int D,
                      int H, int W, int splits /* f16x3_wino_splits: dst = [splits] slabs when > 1 */, const float *in_affine, int in_relu,
                      const float *x_range, hipStream_t s, const int *tile_list /* demand-driven: {count, tile ids} or NULL */, float *gn_part, hipEvent_t t0, hipEvent_t t1);

// conv3d_f16x3_wino_pp.hip: the same kernel contract on the role-split ("ping-pong") schedule; f16x3_wino_launch picks it
void f16x3_wino_pp_launch(dim3 grid, hipStream_t s, hipEvent_t t0, hipEvent_t t1, const float *x, const _Float16 *slabs, const float *hdr,
                          const float *bias, float *dst, int N, int Ci, int Co, int D, int H, int W, int cps, unsigned xb,
                          const float *in_affine, int in_relu, const float *x_range, int tiles, int xcd_on, const int *tile_list,
                          float *gn_part, bool half_products);
int f16x3_wino_pp_saturation(unsigned long long *count, int reset);
// conv3d_f16x3_wino_bt.hip: the same contract (three-product arithmetic only) with one wave per SIMD and a 96 x 128 register tile;
// results bit-identical to the role-split kernel's
void f16x3_wino_bt_launch(dim3 grid, hipStream_t s, hipEvent_t t0, hipEvent_t t1, const float *x, const _Float16 *slabs, const float *hdr,
                          const float *bias, float *dst, int N, int Ci, int Co, int D, int H, int W, int cps, unsigned xb,
                          const float *in_affine, int in_relu, const float *x_range, int tiles, int xcd_on, const int *tile_list,
                          float *gn_part);
int f16x3_wino_bt_saturation(unsigned long long *count, int reset);

// api.hip: the calling thread's conv arithmetic policy (mphip_conv3d_set_half_products): true inside torch.autocast(float16) regions
bool conv_half_products();

// conv3d_bwd_f16x3.hip: 3x3x3 backward-weight on the f16 matrix cores (split precision)
bool bwd_weight_f16x3_supported(int N, int Ci, int Co, int D, int H, int W, int k);
size_t bwd_weight_f16x3_ws_bytes(int N, int Ci, int Co, int D, int H, int W, int k);
int bwd_weight_f16x3_launch(const float *x, const float *x_range, const float *dy, const float *dy_scale, float *dw, int N, int Ci,
                            int Co, int D, int H, int W, int k, void *workspace, hipStream_t s, const int *dy_boxes = nullptr);

// norm.hip: GroupNorm statistics of x [N,C,S] -> stats [N*G][2] (workspace sized by groupnorm_ws_bytes)
// GnTable (optional): the statistics kernels also write what mphip_groupnorm_affine_table would — table[n][c] = (scale, shift) of the
// norm folded into the NEXT conv's staging and the data-independent range descriptor of the normalised tensor — with the same
// arithmetic, saving that launch on the dependent chain (one per residual block).
struct GnTable {
    const float *gamma = nullptr, *beta = nullptr, *w2 = nullptr, *b2 = nullptr;
    float *table = nullptr, *range = nullptr;
    int C = 0, cpg = 0;
    float sqrt_ng = 0.0f;
};
size_t groupnorm_ws_bytes(int N, int C, int S, int G);
int groupnorm_stats_launch(const float *x, float *stats, int N, int C, int S, int G, float eps, void *workspace,
                           hipStream_t s, const GnTable *tbl = nullptr);
// ... from the per-(tile, wave, channel) partial sums (of value - bias) an f16x3 conv launch left in part [C][N*tiles_per_frame][waves][2]
int groupnorm_stats_from_tiles(const float *part, const float *bias, float *stats, int N, int C, int S, int G, float eps, int tiles_per_frame,
                               int waves, hipStream_t s, const GnTable *tbl = nullptr);

}  // namespace mphip
