// One-call entries for the hot slice: mphip_hot_slice_plan_create / _forward / _destroy and mphip_g3d_forward.
//
// In the reference the slice and G3d are single calls (Gbase.forward model.py:1151-1171, G3d.forward model.py:593-597).  Until
// r03 only the per-op C ABI existed and the ~135-launch schedule lived in Python (megaportrait-hack_amd/model.py): every
// consumer had to replay it, and a single frame (B=1) spent more time in Python/ctypes than on the GPU.  A plan owns
//   * the packed weights of every conv of WarpGeneratorS2C / WarpGeneratorC2D / G3d (built from the reference's state-dict
//     tensors, looked up by their state-dict names),
//   * the carve-up of ONE caller-supplied workspace into the step's intermediates (a first-fit arena with explicit
//     lifetimes, replayed identically by the size query),
//   * the two-stream schedule (the C2D generator's latency-bound launches run on the plan's side stream underneath G3d).
// The launches are the SAME kernels in the SAME order as the Python inference path (model.py `_HotSliceRunner._run`): the
// results are bitwise identical, which is how tests/test_gpu_plan.py pins it.  Inference only (training keeps the
// autograd Functions).  Host code; the only kernel here transposes the FlowField 1x1 conv weight.
#include <stdlib.h>
#include <string.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "linspace_tables.h"
#include "mphip_common.h"

using namespace mphip;

namespace {

constexpr float GN_EPS = 1e-5f;                 // nn.GroupNorm default (model.py:506,508,460,309)
constexpr size_t SPLIT_CHAIN_MAX_ELEMS = 1u << 18;   // ops._SPLIT_CHAIN_MAX_ELEMS
constexpr int SPLIT_CHAIN_MAX_SPLITS = 16;            // ops._SPLIT_CHAIN_MAX_SPLITS: from this many slabs on ...
constexpr size_t SPLIT_CHAIN_MAX_SLAB_ELEMS = 1u << 20;   // ... and this many slab elements the ordered reduce (whole chip) beats the split-aware GN kernels
constexpr size_t STATS_SPLIT_MAX_SPAN = 65536;       // ops._STATS_SPLIT_MAX_SPAN
constexpr size_t GN_FUSED_MAX_SPAN = 12288;          // ops._GN_FUSED_MAX_SPAN
constexpr size_t ALIGN = 256;

// ---- arena: first-fit over the caller's workspace, explicit frees, deterministic (the size query replays it dry) -----------
struct Arena {
    struct Block { size_t off, size; bool free; };
    std::vector<Block> blocks;
    char *base = nullptr;
    size_t cap = 0, end = 0, peak = 0;
    bool dry = true, overflow = false;

    void reset(void *b, size_t c, bool d) { blocks.clear(); base = (char *)b; cap = c; end = peak = 0; dry = d; overflow = false; }
    size_t take(size_t bytes) {
        bytes = (bytes + ALIGN - 1) / ALIGN * ALIGN;
        if (bytes == 0) bytes = ALIGN;
        for (size_t i = 0; i < blocks.size(); ++i)
            if (blocks[i].free && blocks[i].size >= bytes) {
                const size_t off = blocks[i].off, rest = blocks[i].size - bytes;
                blocks[i].size = bytes;
                blocks[i].free = false;
                if (rest) blocks.insert(blocks.begin() + i + 1, Block{off + bytes, rest, true});
                return off;
            }
        const size_t off = end;
        blocks.push_back(Block{off, bytes, false});
        end += bytes;
        if (end > peak) peak = end;
        if (!dry && end > cap) overflow = true;
        return off;
    }
    void give(size_t off) {
        for (size_t i = 0; i < blocks.size(); ++i)
            if (blocks[i].off == off && !blocks[i].free) {
                blocks[i].free = true;
                if (i + 1 < blocks.size() && blocks[i + 1].free) { blocks[i].size += blocks[i + 1].size; blocks.erase(blocks.begin() + i + 1); }
                if (i > 0 && blocks[i - 1].free) { blocks[i - 1].size += blocks[i].size; blocks.erase(blocks.begin() + i); --i; }
                if (i + 1 == blocks.size()) { end = blocks[i].off; blocks.pop_back(); }
                return;
            }
    }
    float *ptr(size_t off) const { return (dry || overflow) ? nullptr : (float *)(base + off); }
};

struct Buf {   // an arena allocation (off == SIZE_MAX: not owned / external pointer)
    float *p = nullptr;
    size_t off = (size_t)-1;
};

struct T5 {    // fp32 NCDHW tensor + (optional) range descriptor of its values
    Buf data, range;
    bool has_range = false;   // (explicit: the dry sizing pass has no pointers to look at)
    int n = 0, c = 0, d = 0, h = 0, w = 0;
    size_t numel() const { return (size_t)n * c * d * h * w; }
};

struct ConvW {
    const float *w = nullptr, *b = nullptr;
    int co = 0, ci = 0, k = 0;
    void *pk[2] = {nullptr, nullptr};
    bool fresh[2] = {false, false};
};

struct ConvOut {   // ops.ConvOut: finished tensor (splits == 1) or split-K partial slabs [splits][N,Co,D,H,W] (bias not added)
    T5 t;
    int splits = 1;
    const float *bias = nullptr;
    Buf stats;
    int stats_groups = 0;
    Buf table, table_range;   // the following norm's affine table + range bound, when the conv launch produced them
};

struct Norm { const float *gw = nullptr, *gb = nullptr, *w2 = nullptr, *b2 = nullptr; };   // group_norm.{weight,bias} (+ AdaptiveGroupNorm's)

struct ResBlockAda { ConvW conv1, conv2, res; bool identity = false; Norm n1, n2; };
struct ResBlock { ConvW conv1, conv2, shortcut; bool identity = false; Norm gn1, gn2; };
struct FlowFieldW {
    const float *w1x1 = nullptr, *b1x1 = nullptr;   // conv1x1 [2048,512,1,1]
    float *w1x1_kn = nullptr;                          // [512][2048] (owned)
    float *w_head = nullptr;                           // Gamma @ W^T [512][2048] (owned): (z+e)@Gamma and the 1x1 conv as ONE product
    bool kn_fresh = false;
    ResBlockAda rb[4];
    ConvW conv_out;
    Norm gn;
};
struct Generator { const float *gamma = nullptr; FlowFieldW ff; int invert = 0; };

}  // namespace

struct mphip_hot_slice_plan {
    Generator s2c, c2d;
    ResBlock down[4], up[3];
    ConvW final_conv;
    int C = 96, D = 16, H = 64, W = 64, G = 64;
    bool have_generators = false;
    // index-pipeline tables on the device (captured sizes: built in; others: mphip_hot_slice_plan_set_tables)
    float *lin_d = nullptr, *lin_h = nullptr, *lin_w = nullptr, *aff_base = nullptr;
    bool own_lin[4] = {false, false, false, false};
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int overlap = 1;
    int precision = 1;   // 1 = auto (f16x3 where supported), 0 = exact fp32 everywhere (ops.set_conv_precision)
    int demand = 1;      // G3d's last upsample + conv only where the final warp reads them (include/mphip.h "demand-driven")
    // measurement (bench.py): HIP events around every launch of the dominant conv (96->96 3x3x3 at the plan's full volume) on the
    // stream it is launched on; kind 0 = full launches, 1 = the demand-driven launch
    bool profile = false;
    std::vector<hipEvent_t> prof_ev[2];   // begin/end pairs, in launch order
    size_t prof_used[2] = {0, 0};
    Arena main_arena, side_arena;
    std::unordered_map<int, std::pair<size_t, size_t>> slice_sizes;   // B -> (main, side) arena peaks of the dry pass
    std::vector<void *> owned;   // hipMalloc'ed by the plan
    std::string err;
};

namespace {

using Plan = mphip_hot_slice_plan;

struct Ctx {   // one forward (or one dry sizing pass)
    Plan *p;
    Arena *ar;
    hipStream_t s;
    bool dry;
    int rc = MPHIP_OK;
};

#define RUN(ctx, call)                         \
    do {                                       \
        if (!(ctx).dry && (ctx).rc == MPHIP_OK) { \
            int rc_ = (call);                  \
            if (rc_ != MPHIP_OK) (ctx).rc = rc_; \
        }                                      \
    } while (0)

Buf take(Ctx &c, size_t bytes) {
    Buf b;
    b.off = c.ar->take(bytes);
    b.p = c.ar->ptr(b.off);
    if (!c.dry && c.ar->overflow && c.rc == MPHIP_OK) {
        set_error("hot_slice: workspace too small (need > %zu bytes; query mphip_hot_slice_workspace_bytes)", c.ar->cap);
        c.rc = MPHIP_EWORKSPACE;
    }
    return b;
}
void give(Ctx &c, Buf &b) {
    if (b.off != (size_t)-1) c.ar->give(b.off);
    b = Buf();
}
void give(Ctx &c, T5 &t) { give(c, t.data); give(c, t.range); }
void give(Ctx &c, ConvOut &o) { give(c, o.t); give(c, o.stats); give(c, o.table); give(c, o.table_range); }

T5 new_t5(Ctx &c, int n, int ch, int d, int h, int w, bool with_range) {
    T5 t;
    t.n = n; t.c = ch; t.d = d; t.h = h; t.w = w;
    t.data = take(c, t.numel() * sizeof(float));
    if (with_range) t.range = take(c, MPHIP_RANGE_FLOATS * sizeof(float));
    t.has_range = with_range;
    return t;
}

__global__ void __launch_bounds__(256) transpose_kernel(const float *__restrict__ in, float *__restrict__ out, int rows, int cols) {
    // out[c][r] = in[r][c]; 32x32 tiles through LDS (the FlowField 1x1 conv weight [2048][512] -> [512][2048], once per weight version)
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (by + i < rows && bx + tx < cols) tile[i][tx] = in[(size_t)(by + i) * cols + bx + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (bx + i < cols && by + tx < rows) out[(size_t)(bx + i) * rows + by + tx] = tile[tx][i];
}

int precision_for(const Ctx &c, int n, int ci, int co, int d, int h, int w, int k) {
    if (c.p->precision == 0) return 0;                                  // exact fp32 MFMA everywhere
    return mphip_conv3d_supported(n, ci, co, d, h, w, k, 1) ? 1 : 0;   // "auto": f16x3 wherever the kernel covers the shape
}

// packed weights: allocated on first use of a precision, (re)packed on the forward's stream when the plan was refreshed
const void *packed(Ctx &c, ConvW &cw, int prec) {
    if (c.dry) return nullptr;
    if (!cw.pk[prec]) {
        const size_t bytes = mphip_packed_weight_bytes(cw.co, cw.ci, cw.k, prec);
        void *q = nullptr;
        if (bytes == 0 || hipMalloc(&q, bytes) != hipSuccess) {
            set_error("hot_slice: cannot allocate %zu bytes of packed weights (Co=%d Ci=%d k=%d precision %d)", bytes, cw.co, cw.ci, cw.k, prec);
            c.rc = MPHIP_ELAUNCH;
            return nullptr;
        }
        c.p->owned.push_back(q);
        cw.pk[prec] = q;
        cw.fresh[prec] = false;
    }
    if (!cw.fresh[prec] && !c.dry && c.rc == MPHIP_OK) {   // ("fresh" only once the pack was really issued — a forward that failed
        RUN(c, mphip_pack_conv_weight(cw.w, cw.pk[prec], cw.co, cw.ci, cw.k, prec, c.s));   // earlier must not leave a never-written buffer marked usable)
        if (c.rc == MPHIP_OK) cw.fresh[prec] = true;
    }
    return cw.pk[prec];
}

// measurement: bracket the NEXT conv launch (the conv kernel itself) with a pair of events of the given kind
void prof_arm(Ctx &c, const ConvW &cw, int d, int h, int w, int kind) {
    Plan *p = c.p;
    if (c.dry || c.rc != MPHIP_OK /* the launch this would time is skipped (RUN): do not arm */ || !p->profile || cw.ci != 96 || cw.co != 96 || cw.k != 3 || d != p->D || h != p->H || w != p->W) return;
    if (p->prof_used[kind] + 2 > p->prof_ev[kind].size())
        for (int i = 0; i < 2; ++i) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            p->prof_ev[kind].push_back(e);
        }
    mphip_conv3d_time_next_launch(p->prof_ev[kind][p->prof_used[kind]], p->prof_ev[kind][p->prof_used[kind] + 1]);
    p->prof_used[kind] += 2;
}

const float *range_for(Ctx &c, T5 &x) {   // ops._range_for: the producer's descriptor, else one streaming pass
    if (!x.has_range) {
        x.range = take(c, MPHIP_RANGE_FLOATS * sizeof(float));
        x.has_range = true;
        RUN(c, mphip_absmax_range(x.data.p, x.numel(), x.range.p, c.s));
    }
    return x.range.p;
}

bool gn_in_conv_ok(const Ctx &c, int n, int d, int h, int w, const ConvW &pc2);

// ops.conv3d: finished tensor (+ the statistics of the GroupNorm that follows when gn_groups; + that norm's affine table when it
// will be folded into `next`'s staging: nm / next given)
ConvOut conv3d(Ctx &c, T5 &x, ConvW &cw, int gn_groups, const Norm *nm = nullptr, const ConvW *next = nullptr) {
    const int n = x.n, d = x.d, h = x.h, w = x.w;
    const int prec = precision_for(c, n, cw.ci, cw.co, d, h, w, cw.k);
    const void *wp = packed(c, cw, prec);
    const float *xr = prec == 1 ? range_for(c, x) : nullptr;
    const size_t ws_bytes = gn_groups ? mphip_conv3d_gn_workspace_bytes(n, cw.ci, cw.co, d, h, w, cw.k, prec, gn_groups)
                                      : mphip_conv3d_workspace_bytes(n, cw.ci, cw.co, d, h, w, cw.k, prec);
    Buf ws;
    if (ws_bytes) ws = take(c, ws_bytes);
    ConvOut o;
    o.t = new_t5(c, n, cw.co, d, h, w, false);
    prof_arm(c, cw, d, h, w, 0);
    if (gn_groups) {
        o.stats = take(c, (size_t)n * gn_groups * 2 * sizeof(float));
        o.stats_groups = gn_groups;
        if (nm && next && gn_in_conv_ok(c, n, d, h, w, *next)) {
            o.table = take(c, (size_t)n * cw.co * 2 * sizeof(float));
            o.table_range = take(c, MPHIP_RANGE_FLOATS * sizeof(float));
            RUN(c, mphip_conv3d_gn_table_fwd(x.data.p, xr, wp, cw.b, o.t.data.p, o.stats.p, nm->gw, nm->gb, nm->w2, nm->b2, o.table.p, o.table_range.p, n,
                                             cw.ci, cw.co, d, h, w, cw.k, prec, gn_groups, GN_EPS, ws.p, ws_bytes, c.s));
        } else {
            RUN(c, mphip_conv3d_gn_fwd(x.data.p, xr, wp, cw.b, o.t.data.p, o.stats.p, n, cw.ci, cw.co, d, h, w, cw.k, prec, gn_groups, GN_EPS,
                                       ws.p, ws_bytes, c.s));
        }
    } else {
        RUN(c, mphip_conv3d_fwd(x.data.p, xr, wp, cw.b, o.t.data.p, n, cw.ci, cw.co, d, h, w, cw.k, prec, ws.p, ws_bytes, c.s));
    }
    give(c, ws);
    return o;
}

// ops.conv3d_split: split-K slabs are kept for the GroupNorm kernels when the tensor is small
ConvOut conv3d_split(Ctx &c, T5 &x, ConvW &cw, int gn_groups, const Norm *nm = nullptr, const ConvW *next = nullptr) {
    const int n = x.n, d = x.d, h = x.h, w = x.w;
    const int prec = precision_for(c, n, cw.ci, cw.co, d, h, w, cw.k);
    const int splits = mphip_conv3d_splits(n, cw.ci, cw.co, d, h, w, cw.k, prec);
    const size_t elems = (size_t)n * cw.co * d * h * w;
    if (splits > 1 && (elems > SPLIT_CHAIN_MAX_ELEMS || (splits >= SPLIT_CHAIN_MAX_SPLITS && (size_t)splits * elems > SPLIT_CHAIN_MAX_SLAB_ELEMS)))
        return conv3d(c, x, cw, gn_groups, nm, next);
    if (gn_groups && splits == 1 && prec == 1) return conv3d(c, x, cw, gn_groups, nm, next);
    const void *wp = packed(c, cw, prec);
    const float *xr = prec == 1 ? range_for(c, x) : nullptr;
    ConvOut o;
    o.t.n = n; o.t.c = cw.co; o.t.d = d; o.t.h = h; o.t.w = w;
    o.t.data = take(c, (size_t)splits * elems * sizeof(float));
    o.splits = splits;
    o.bias = splits > 1 ? cw.b : nullptr;
    RUN(c, mphip_conv3d_fwd_split(x.data.p, xr, wp, cw.b, o.t.data.p, n, cw.ci, cw.co, d, h, w, cw.k, prec, nullptr, 0, c.s));
    return o;
}

// ops.groupnorm_stats: (mean, rstd) of a ConvOut; reuses the statistics a conv launch already produced
void ensure_stats(Ctx &c, ConvOut &y, int groups) {
    if (y.stats.off != (size_t)-1 && y.stats_groups == groups) return;   // produced by the conv launch itself
    const int n = y.t.n, ch = y.t.c, s = y.t.d * y.t.h * y.t.w;
    if (y.stats.off != (size_t)-1) give(c, y.stats);
    y.stats = take(c, (size_t)n * groups * 2 * sizeof(float));
    y.stats_groups = groups;
    if (y.splits > 1) {
        if ((size_t)(ch / groups) * s > STATS_SPLIT_MAX_SPAN) {
            if (c.rc == MPHIP_OK && !c.dry) {
                set_error("hot_slice: split-K GroupNorm span %zu floats > %zu (shape outside the planned schedule)", (size_t)(ch / groups) * s,
                          STATS_SPLIT_MAX_SPAN);
                c.rc = MPHIP_EINVAL;
            }
            return;
        }
        RUN(c, mphip_groupnorm_stats_split(y.t.data.p, y.splits, y.bias, y.stats.p, n, ch, s, groups, GN_EPS, c.s));
        return;
    }
    const size_t ws_bytes = mphip_groupnorm_workspace_bytes(n, ch, s, groups);
    Buf ws = take(c, ws_bytes);
    RUN(c, mphip_groupnorm_stats(y.t.data.p, y.stats.p, n, ch, s, groups, GN_EPS, ws.p, ws_bytes, c.s));
    give(c, ws);
}

// ops.groupnorm_apply (+second affine, +residual, +ReLU, +tanh, then 2x2x2 pool or nearest upsample); consumes nothing
T5 groupnorm_apply(Ctx &c, ConvOut &x, const Norm &nm, int groups, const ConvOut *res, bool relu, bool tanh_, bool pool2, int ud, int uh, int uw) {
    const int n = x.t.n, ch = x.t.c, d = x.t.d, h = x.t.h, w = x.t.w;
    const int rs = res ? res->splits : 1;
    const bool general = x.splits > 1 || rs > 1 || ud != 1 || uh != 1 || uw != 1;
    T5 y = pool2 ? new_t5(c, n, ch, d / 2, h / 2, w / 2, true) : new_t5(c, n, ch, d * ud, h * uh, w * uw, true);
    if (general)
        RUN(c, mphip_groupnorm_apply_split(x.t.data.p, x.splits, x.bias, x.stats.p, nm.gw, nm.gb, nm.w2, nm.b2, res ? res->t.data.p : nullptr, rs,
                                           res ? res->bias : nullptr, y.data.p, y.range.p, n, ch, d, h, w, groups, relu, tanh_, pool2, ud, uh, uw,
                                           c.s));
    else
        RUN(c, mphip_groupnorm_apply(x.t.data.p, x.stats.p, nm.gw, nm.gb, nm.w2, nm.b2, res ? res->t.data.p : nullptr, y.data.p, y.range.p, n, ch,
                                     d, h, w, groups, relu, tanh_, pool2, c.s));
    return y;
}

bool groupnorm_fused_ok(const ConvOut &x, int groups) {
    return (size_t)(x.t.c / groups) * x.t.d * x.t.h * x.t.w <= GN_FUSED_MAX_SPAN;
}

// ops.groupnorm_small: statistics + apply in one launch (FlowField)
T5 groupnorm_small(Ctx &c, ConvOut &x, const Norm &nm, int groups, const ConvOut *res, bool relu, bool tanh_, int ud, int uh, int uw) {
    const int n = x.t.n, ch = x.t.c, d = x.t.d, h = x.t.h, w = x.t.w;
    T5 y = new_t5(c, n, ch, d * ud, h * uh, w * uw, false);
    RUN(c, mphip_groupnorm_small_fused(x.t.data.p, x.splits, x.bias, nm.gw, nm.gb, nm.w2, nm.b2, res ? res->t.data.p : nullptr, res ? res->splits : 1,
                                       res ? res->bias : nullptr, y.data.p, nullptr, n, ch, d, h, w, groups, GN_EPS, relu, tanh_, ud, uh, uw, c.s));
    return y;
}

bool gn_in_conv_ok(const Ctx &c, int n, int d, int h, int w, const ConvW &pc2) {
    if (c.p->precision != 1 || pc2.k != 3 || pc2.ci > 768) return false;
    return mphip_conv3d_supported(n, pc2.ci, pc2.co, d, h, w, pc2.k, 1) != 0;
}
bool gn_in_conv_ok(const Ctx &c, const ConvOut &y, const ConvW &pc2) { return gn_in_conv_ok(c, y.t.n, y.t.d, y.t.h, y.t.w, pc2); }

// ops.conv3d_gn_in: conv(relu(GN(x))) with the norm folded into the conv's staging; also the statistics of its own output
ConvOut conv3d_gn_in(Ctx &c, ConvOut &y, const Norm &nm, int groups, ConvW &pc2, int out_gn_groups) {
    const int n = y.t.n, ci = y.t.c, d = y.t.d, h = y.t.h, w = y.t.w;
    const bool have = y.table.off != (size_t)-1;   // the producing conv launch already wrote the table and the bound
    Buf table = have ? y.table : take(c, (size_t)n * ci * 2 * sizeof(float));
    Buf xr = have ? y.table_range : take(c, MPHIP_RANGE_FLOATS * sizeof(float));
    y.table = Buf();
    y.table_range = Buf();
    if (!have) RUN(c, mphip_groupnorm_affine_table(y.stats.p, nm.gw, nm.gb, nm.w2, nm.b2, table.p, xr.p, n, ci, d * h * w, groups, c.s));
    const void *wp = packed(c, pc2, 1);
    const size_t ws_bytes = mphip_conv3d_gn_workspace_bytes(n, ci, pc2.co, d, h, w, pc2.k, 1, out_gn_groups);
    Buf ws = take(c, ws_bytes);
    ConvOut o;
    o.t = new_t5(c, n, pc2.co, d, h, w, false);
    o.stats = take(c, (size_t)n * out_gn_groups * 2 * sizeof(float));
    o.stats_groups = out_gn_groups;
    prof_arm(c, pc2, d, h, w, 0);
    RUN(c, mphip_conv3d_gnin_gn_fwd(y.t.data.p, table.p, xr.p, 1, wp, pc2.b, o.t.data.p, o.stats.p, n, ci, pc2.co, d, h, w, pc2.k, 1, out_gn_groups,
                                    GN_EPS, ws.p, ws_bytes, c.s));
    give(c, ws);
    give(c, table);
    give(c, xr);
    return o;
}

ConvOut as_convout(const T5 &x) {   // a plain tensor in ConvOut clothing (not owned: the caller keeps x)
    ConvOut o;
    o.t = x;
    o.t.data.off = (size_t)-1;
    o.t.range.off = (size_t)-1;
    return o;   // (give() ignores it)
}

// The generator chains are written over L "lanes" (L = 1: one generator on one stream; L = 2: WarpGeneratorS2C on the caller's
// stream and WarpGeneratorC2D on the side stream, advanced in LOCKSTEP — every step is enqueued for lane 0, then for lane 1).
// Lockstep is a host-order requirement: the ROCm runtime holds launches that wait on a not-yet-complete cross-stream event — and
// everything issued after them — in software, and feeds them to the hardware queues one by one (~6 us each) once the event
// completes.  Issued one chain after the other, ~30 held launches kept the second chain out of its queue for 160-230 us at the
// start of every step (profiles/r03_timeline_plan_sidefirst.txt); interleaved, both ~0.3 ms chains of tiny latency-bound
// kernels run side by side from the first microsecond and are done before G3d's first conv takes every register of the chip.
struct GenLane {
    Ctx *c;
    Generator *g;
    const float *R, *t, *z, *e;
};

// ResBlock3D_Adaptive._forward, inference branch (model.py:369-408); consumes x
template <int L>
void resblock_ada(GenLane (&ln)[L], int blk, T5 (&x)[L], int ud, int uh, int uw) {
    ConvOut y[L], y2[L], res[L];
    T5 a[L], out[L];
    ResBlockAda *b[L];
    for (int l = 0; l < L; ++l) b[l] = &ln[l].g->ff.rb[blk];
    static const bool ff_fused = !(getenv("MPHIP_FF_FUSED") && getenv("MPHIP_FF_FUSED")[0] == '0');   // dev: same-box A/B (ops._FF_FUSED)
    if (ff_fused && !b[0]->identity &&
        mphip_flowfield_conv_gn_supported(b[0]->conv1.ci, b[0]->conv1.co, x[0].d, x[0].h, x[0].w, 0, 32) &&
        mphip_flowfield_conv_gn_supported(b[0]->conv2.ci, b[0]->conv2.co, x[0].d, x[0].h, x[0].w, b[0]->res.ci, 32)) {
        // FlowField's levels: each half of the block is ONE launch (csrc/flowfield.hip; model.ResBlock3D_Adaptive._forward does the same)
        // level 1 (4x1x1): from the compact copy of the three usable taps (flowfield.hip; kept in the conv's precision-1 pack slot, which these
        // fp32 layers never use, and refreshed with the other packs)
        auto compact = [&](Ctx &c, ConvW &cw, int d, int h, int w) -> const void * {
            const size_t bytes = mphip_flowfield_compact_weight_bytes(cw.ci, cw.co, d, h, w);
            if (!bytes || c.dry) return nullptr;
            if (!cw.pk[1]) {
                void *q = nullptr;
                if (hipMalloc(&q, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }   // (falls back to the original tensor)
                c.p->owned.push_back(q);
                cw.pk[1] = q;
                cw.fresh[1] = false;
            }
            if (!cw.fresh[1] && c.rc == MPHIP_OK) {   // (fresh only when the copy was issued and succeeded, see packed())
                RUN(c, mphip_flowfield_compact_weight(cw.w, cw.pk[1], cw.ci, cw.co, c.s));
                if (c.rc == MPHIP_OK) cw.fresh[1] = true;
            }
            return cw.pk[1];
        };
        for (int l = 0; l < L; ++l) {
            Ctx &c = *ln[l].c;
            a[l] = new_t5(c, x[l].n, b[l]->conv1.co, x[l].d, x[l].h, x[l].w, false);
            if (const void *wc = compact(c, b[l]->conv1, x[l].d, x[l].h, x[l].w))
                RUN(c, mphip_flowfield_conv_gn_compact(x[l].data.p, wc, b[l]->conv1.b, b[l]->n1.gw, b[l]->n1.gb, b[l]->n1.w2, b[l]->n1.b2, nullptr,
                                                       nullptr, nullptr, a[l].data.p, x[l].n, b[l]->conv1.ci, b[l]->conv1.co, x[l].d, x[l].h, x[l].w, 0, 1, 1,
                                                       1, 32, GN_EPS, 1, c.s));
            else
                RUN(c, mphip_flowfield_conv_gn(x[l].data.p, b[l]->conv1.w, b[l]->conv1.b, b[l]->n1.gw, b[l]->n1.gb, b[l]->n1.w2, b[l]->n1.b2, nullptr,
                                               nullptr, nullptr, a[l].data.p, x[l].n, b[l]->conv1.ci, b[l]->conv1.co, x[l].d, x[l].h, x[l].w, 0, 1, 1, 1,
                                               32, GN_EPS, 1, c.s));
        }
        for (int l = 0; l < L; ++l) {
            Ctx &c = *ln[l].c;
            out[l] = new_t5(c, x[l].n, b[l]->conv2.co, x[l].d * ud, x[l].h * uh, x[l].w * uw, false);
            if (const void *wc = compact(c, b[l]->conv2, x[l].d, x[l].h, x[l].w))
                RUN(c, mphip_flowfield_conv_gn_compact(a[l].data.p, wc, b[l]->conv2.b, b[l]->n2.gw, b[l]->n2.gb, b[l]->n2.w2, b[l]->n2.b2,
                                                       x[l].data.p, b[l]->res.w, b[l]->res.b, out[l].data.p, x[l].n, b[l]->conv2.ci, b[l]->conv2.co, x[l].d,
                                                       x[l].h, x[l].w, b[l]->res.ci, ud, uh, uw, 32, GN_EPS, 1, c.s));
            else
                RUN(c, mphip_flowfield_conv_gn(a[l].data.p, b[l]->conv2.w, b[l]->conv2.b, b[l]->n2.gw, b[l]->n2.gb, b[l]->n2.w2, b[l]->n2.b2,
                                               x[l].data.p, b[l]->res.w, b[l]->res.b, out[l].data.p, x[l].n, b[l]->conv2.ci, b[l]->conv2.co, x[l].d,
                                               x[l].h, x[l].w, b[l]->res.ci, ud, uh, uw, 32, GN_EPS, 1, c.s));
        }
        for (int l = 0; l < L; ++l) {
            give(*ln[l].c, a[l]);
            give(*ln[l].c, x[l]);
            x[l] = out[l];
        }
        return;
    }
    for (int l = 0; l < L; ++l) y[l] = conv3d_split(*ln[l].c, x[l], b[l]->conv1, 32, &b[l]->n1, &b[l]->conv2);
    const bool tiny = groupnorm_fused_ok(y[0], 32);   // (shape decisions are the same on every lane: same batch, same layer)
    if (tiny) {
        for (int l = 0; l < L; ++l) { a[l] = groupnorm_small(*ln[l].c, y[l], b[l]->n1, 32, nullptr, true, false, 1, 1, 1); give(*ln[l].c, y[l]); }
        for (int l = 0; l < L; ++l) { y2[l] = conv3d_split(*ln[l].c, a[l], b[l]->conv2, 0); give(*ln[l].c, a[l]); }
    } else {
        for (int l = 0; l < L; ++l) ensure_stats(*ln[l].c, y[l], 32);
        if (y[0].splits == 1 && gn_in_conv_ok(*ln[0].c, y[0], b[0]->conv2)) {
            for (int l = 0; l < L; ++l) { y2[l] = conv3d_gn_in(*ln[l].c, y[l], b[l]->n1, 32, b[l]->conv2, 32); give(*ln[l].c, y[l]); }
        } else {
            for (int l = 0; l < L; ++l) { a[l] = groupnorm_apply(*ln[l].c, y[l], b[l]->n1, 32, nullptr, true, false, false, 1, 1, 1); give(*ln[l].c, y[l]); }
            for (int l = 0; l < L; ++l) { y2[l] = conv3d_split(*ln[l].c, a[l], b[l]->conv2, 32); give(*ln[l].c, a[l]); }
        }
    }
    for (int l = 0; l < L; ++l) res[l] = b[l]->identity ? as_convout(x[l]) : conv3d_split(*ln[l].c, x[l], b[l]->res, 0);
    if (tiny) {
        for (int l = 0; l < L; ++l) out[l] = groupnorm_small(*ln[l].c, y2[l], b[l]->n2, 32, &res[l], true, false, ud, uh, uw);
    } else {
        for (int l = 0; l < L; ++l) ensure_stats(*ln[l].c, y2[l], 32);
        for (int l = 0; l < L; ++l) out[l] = groupnorm_apply(*ln[l].c, y2[l], b[l]->n2, 32, &res[l], true, false, false, ud, uh, uw);
    }
    for (int l = 0; l < L; ++l) {
        give(*ln[l].c, y2[l]);
        if (!b[l]->identity) give(*ln[l].c, res[l]);
        give(*ln[l].c, x[l]);
        x[l] = out[l];
    }
}

// _WarpGenerator.forward (model.py:927-1024: FlowField((z+e) @ Gamma), compute_rt_warp, compose) followed by the coordinate pass of
// the warp its field feeds (model.py:1036-1058) -> clipped sample positions [B,D,H,W,3].  When the volume's H, W equal the
// field's grid (64: the reference's size) the field itself is never written: compose + resize + coordinate chain run as one
// kernel on the 2*D planes the resize touches (mphip_warp_field_coords).
template <int L>
void generator_coords(GenLane (&ln)[L], int B, Buf (&coords)[L]) {
    static const int UPS[4][3] = {{2, 2, 2}, {2, 2, 2}, {1, 2, 2}, {1, 2, 2}};
    Plan *p = ln[0].c->p;
    const int G = p->G;
    T5 x[L], em[L];
    ConvOut y[L];
    Buf theta[L];
    for (int l = 0; l < L; ++l) {
        Ctx &c = *ln[l].c;
        FlowFieldW &ff = ln[l].g->ff;
        if (!c.dry && !ff.kn_fresh) {
            // conv1x1.weight [2048][512] -> [K=512][N=2048], then Gamma @ that: (z+e)@Gamma followed by the 1x1 conv is one linear map
            hipLaunchKernelGGL(transpose_kernel, dim3(512 / 32, 2048 / 32), dim3(256), 0, c.s, ff.w1x1, ff.w1x1_kn, 2048, 512);
            if (c.rc == MPHIP_OK) c.rc = check_launch("hot_slice(transpose conv1x1)");
            RUN(c, mphip_small_gemm(ln[l].g->gamma, nullptr, ff.w1x1_kn, nullptr, ff.w_head, 512, 2048, 512, 512, 1, 2048, 1, c.s));
            ff.kn_fresh = true;
        }
        x[l] = new_t5(c, B, 512, 4, 1, 1, false);   // [B,2048] viewed as [B,512,4,1,1] (model.py:425)
        RUN(c, mphip_add_matmul(ln[l].z, ln[l].e, ff.w_head, ff.b1x1, x[l].data.p, B, 512, 2048, 0, c.s));
    }
    for (int i = 0; i < 4; ++i) resblock_ada<L>(ln, i, x, UPS[i][0], UPS[i][1], UPS[i][2]);
    static const bool ff_fused = !(getenv("MPHIP_FF_FUSED") && getenv("MPHIP_FF_FUSED")[0] == '0');
    const ConvW &co0 = ln[0].g->ff.conv_out;
    const bool out_fused = ff_fused && co0.ci == 32 && co0.co == 3 && co0.k == 3 && x[0].d == 16 && x[0].h == 16 && x[0].w == 16;
    if (out_fused) {   // the output head as a direct 3-channel conv + one normalising pass (csrc/flowfield.hip; ops.flowfield_out)
        for (int l = 0; l < L; ++l) {
            Ctx &c = *ln[l].c;
            FlowFieldW &ff = ln[l].g->ff;
            em[l] = new_t5(c, B, 3, 16, 16, 16, false);
            const size_t wsb = mphip_flowfield_out_workspace_bytes(B);
            Buf ws = take(c, wsb);
            RUN(c, mphip_flowfield_out(x[l].data.p, ff.conv_out.w, ff.conv_out.b, ff.gn.gw, ff.gn.gb, em[l].data.p, B, GN_EPS, ws.p, wsb, c.s));
            give(c, ws);
            give(c, x[l]);
        }
    } else {
    for (int l = 0; l < L; ++l) { y[l] = conv3d_split(*ln[l].c, x[l], ln[l].g->ff.conv_out, 0); give(*ln[l].c, x[l]); }
    if (groupnorm_fused_ok(y[0], 1)) {
        for (int l = 0; l < L; ++l) em[l] = groupnorm_small(*ln[l].c, y[l], ln[l].g->ff.gn, 1, nullptr, true, true, 1, 1, 1);
    } else {
        for (int l = 0; l < L; ++l) ensure_stats(*ln[l].c, y[l], 1);
        for (int l = 0; l < L; ++l) em[l] = groupnorm_apply(*ln[l].c, y[l], ln[l].g->ff.gn, 1, nullptr, true, true, false, 1, 1, 1);
    }
    for (int l = 0; l < L; ++l) give(*ln[l].c, y[l]);
    }
    for (int l = 0; l < L; ++l) {
        Ctx &c = *ln[l].c;
        theta[l] = take(c, (size_t)B * 12 * sizeof(float));
        RUN(c, mphip_rt_theta(ln[l].R, ln[l].t, theta[l].p, B, ln[l].g->invert, c.s));
    }
    for (int l = 0; l < L; ++l) {
        Ctx &c = *ln[l].c;
        coords[l] = take(c, (size_t)B * p->D * p->H * p->W * 3 * sizeof(float));
        if (p->H == G && p->W == G) {
            RUN(c, mphip_warp_field_coords(theta[l].p, em[l].data.p, p->aff_base, p->lin_d, p->lin_h, p->lin_w, coords[l].p, B, em[l].d, em[l].h, em[l].w,
                                           G, p->D, c.s));
        } else {
            T5 wf = new_t5(c, B, 3, G, G, G, false);
            RUN(c, mphip_warp_field_compose(theta[l].p, em[l].data.p, p->aff_base, wf.data.p, nullptr, nullptr, B, em[l].d, em[l].h, em[l].w, G, c.s));
            RUN(c, mphip_warp_coords(wf.data.p, p->lin_d, p->lin_h, p->lin_w, coords[l].p, B, p->D, p->H, p->W, G, G, G, c.s));
            give(c, wf);
        }
        give(c, theta[l]);
        give(c, em[l]);
    }
}

// ResBlock3D._forward, inference branch (model.py:500-528); consumes x.  `hook`: called right after conv1 was launched.
template <typename Hook>
T5 resblock(Ctx &c, ResBlock &b, T5 &x, bool pool_after, Hook hook) {
    ConvOut identity = b.identity ? as_convout(x) : conv3d_split(c, x, b.shortcut, 0);
    ConvOut y = conv3d_split(c, x, b.conv1, 32, &b.gn1, &b.conv2);   // (+ gn1's affine table when it will be folded into conv2)
    hook();
    ensure_stats(c, y, 32);
    ConvOut y2;
    if (y.splits == 1 && gn_in_conv_ok(c, y, b.conv2)) {
        y2 = conv3d_gn_in(c, y, b.gn1, 32, b.conv2, 32);   // GN1 + ReLU folded into conv2's input staging
        give(c, y);
    } else {
        T5 a = groupnorm_apply(c, y, b.gn1, 32, nullptr, true, false, false, 1, 1, 1);
        give(c, y);
        y2 = conv3d_split(c, a, b.conv2, 32);
        give(c, a);
    }
    ensure_stats(c, y2, 32);
    T5 out = groupnorm_apply(c, y2, b.gn2, 32, &identity, true, false, pool_after, 1, 1, 1);
    give(c, y2);
    if (!b.identity) give(c, identity);
    give(c, x);
    return out;
}

struct Roi {   // demand-driven tail of G3d: the boxes of voxels the final warp reads (device) and the conv kernel's tile
    const int *box = nullptr;
    bool on = false;
    int tile[3] = {0, 0, 0};
};

T5 upsample2(Ctx &c, T5 &x, const Roi *roi = nullptr) {   // nn.Upsample(scale_factor=2, trilinear, align_corners=True), model.py:585-589; consumes x
    T5 y;
    y.n = x.n; y.c = x.c; y.d = 2 * x.d; y.h = 2 * x.h; y.w = 2 * x.w;
    y.data = take(c, y.numel() * sizeof(float));
    if (roi && roi->on)
        RUN(c, mphip_upsample_trilinear2_roi(x.data.p, y.data.p, roi->box, 0, x.n, x.c, x.d, x.h, x.w, roi->tile[0], roi->tile[1], roi->tile[2], c.s));
    else
        RUN(c, mphip_upsample_trilinear2(x.data.p, y.data.p, x.n * x.c, x.d, x.h, x.w, c.s));
    y.range = x.range;   // convex combinations of x: the descriptor carries over (ops.upsample_trilinear2)
    y.has_range = x.has_range;
    x.range = Buf();
    x.has_range = false;
    give(c, x.data);
    return y;
}

// G3d.forward (model.py:571-597); consumes x.  out: caller buffer (may be nullptr: arena)
// `tail(n)`: called before the last upsample; returns the sample boxes of the warp that will read the result (or an "off" Roi)
template <typename Hook, typename Tail>
T5 g3d(Ctx &c, T5 &x, bool external_out, float *out, Hook hook, Tail tail) {
    Plan *p = c.p;
    auto none = [] {};
    T5 t = resblock(c, p->down[0], x, true, hook);
    t = resblock(c, p->down[1], t, true, none);
    t = resblock(c, p->down[2], t, true, none);
    t = resblock(c, p->down[3], t, false, none);
    t = resblock(c, p->up[0], t, false, none);
    t = upsample2(c, t);
    t = resblock(c, p->up[1], t, false, none);
    t = upsample2(c, t);
    t = resblock(c, p->up[2], t, false, none);
    ConvW &cw = p->final_conv;
    Roi roi = tail();
    if (roi.on) roi.on = mphip_conv3d_roi_granule(t.n, cw.ci, cw.co, 2 * t.d, 2 * t.h, 2 * t.w, cw.k, precision_for(c, t.n, cw.ci, cw.co, 2 * t.d, 2 * t.h, 2 * t.w, cw.k), roi.tile) != 0;
    t = upsample2(c, t, &roi);
    // final_conv: ops.conv3d (finished tensor); demand-driven: only the tiles the final warp reads
    const int prec = precision_for(c, t.n, cw.ci, cw.co, t.d, t.h, t.w, cw.k);
    const void *wp = packed(c, cw, prec);
    const float *xr = prec == 1 ? range_for(c, t) : nullptr;
    const size_t ws_bytes = roi.on ? mphip_conv3d_roi_workspace_bytes(t.n, cw.ci, cw.co, t.d, t.h, t.w, cw.k, prec)
                                   : mphip_conv3d_workspace_bytes(t.n, cw.ci, cw.co, t.d, t.h, t.w, cw.k, prec);
    Buf ws;
    if (ws_bytes) ws = take(c, ws_bytes);
    T5 y;
    y.n = t.n; y.c = cw.co; y.d = t.d; y.h = t.h; y.w = t.w;
    if (external_out) y.data.p = out; else y.data = take(c, y.numel() * sizeof(float));
    prof_arm(c, cw, t.d, t.h, t.w, roi.on ? 1 : 0);
    if (roi.on)
        RUN(c, mphip_conv3d_fwd_roi(t.data.p, xr, wp, cw.b, y.data.p, roi.box, 0, t.n, cw.ci, cw.co, t.d, t.h, t.w, cw.k, prec, ws.p, ws_bytes, c.s));
    else
        RUN(c, mphip_conv3d_fwd(t.data.p, xr, wp, cw.b, y.data.p, t.n, cw.ci, cw.co, t.d, t.h, t.w, cw.k, prec, ws.p, ws_bytes, c.s));
    give(c, ws);
    give(c, t);
    return y;
}

// _HotSliceRunner._run for <= 64 frames (model.py:1151-1171)
int run_slice(Plan *p, const float *vs, const float *es, const float *Rs, const float *ts, const float *zs, const float *Rd, const float *td,
              const float *zd, float *out, int B, void *workspace, size_t workspace_bytes, hipStream_t s, bool dry, size_t *need) {
    const bool overlap = p->overlap && !dry;
    // the side stream's arena sits behind the main one: sizes from the dry pass
    size_t side_bytes = 0, main_bytes = 0;
    if (!dry) {
        auto hit = p->slice_sizes.find(B);
        if (hit == p->slice_sizes.end()) {
            run_slice(p, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, B, nullptr, 0, nullptr, true, nullptr);
            hit = p->slice_sizes.emplace(B, std::make_pair(p->main_arena.peak, p->side_arena.peak)).first;
        }
        main_bytes = hit->second.first;
        side_bytes = hit->second.second;
        if (workspace_bytes < main_bytes + side_bytes || !workspace) {
            set_error("hot_slice_forward: workspace %zu bytes < required %zu", workspace_bytes, main_bytes + side_bytes);
            return MPHIP_EWORKSPACE;
        }
    }
    p->main_arena.reset(workspace, dry ? 0 : main_bytes, dry);
    p->side_arena.reset(dry ? nullptr : (char *)workspace + main_bytes, dry ? 0 : side_bytes, dry);
    Ctx cm{p, &p->main_arena, s, dry};
    Ctx cs{p, &p->side_arena, overlap ? p->side : s, dry};
    // fork: inputs produced on the caller's stream are visible to the side stream.  The event is RECORDED here; the side stream's
    // wait on it is issued later (fork_wait), after the caller-stream launches that should not queue up behind it (see GenLane)
    static const int order = getenv("MPHIP_PLAN_CHAIN_ORDER") ? atoi(getenv("MPHIP_PLAN_CHAIN_ORDER")) : 0;   // dev A/B: 0 lockstep (default), 1 S2C chain first, then wait + C2D chain (measured equal within 0.5 %)
    bool fork_waited = false;
    auto fork_wait = [&]() -> bool {
        if (!overlap || fork_waited) return true;
        fork_waited = true;
        if (hipStreamWaitEvent(p->side, p->ev_fork, 0) != hipSuccess) {
            set_error("hot_slice_forward: stream fork failed: %s", hipGetErrorString(hipGetLastError()));
            return false;
        }
        return true;
    };
    if (overlap && hipEventRecord(p->ev_fork, s) != hipSuccess) {
        set_error("hot_slice_forward: stream fork failed: %s", hipGetErrorString(hipGetLastError()));
        return MPHIP_ELAUNCH;
    }
    // the two generators.  Demand-driven tail: the C2D boxes must exist before G3d's LAST upsample, so the C2D chain cannot hide
    // under final_conv any more (G3d's persistent conv workgroups own every register of a CU: a side-stream kernel only runs in
    // the gaps between conv launches) — it runs beside the equally latency-bound S2C chain, in lockstep (see GenLane).  With the
    // full tail the old order stands: critical path first, the C2D generator's ~25 launches behind G3d's first conv.
    // K2's corner image (warp.hip): vs is an input, so the copy runs here, beside the generators' latency-bound chains, not in front of K2
    Buf cimg = take(cm, mphip_warp_corner_image_bytes(B, p->C));
    RUN(cm, mphip_warp_corner_image(vs, cimg.p, mphip_warp_corner_image_bytes(B, p->C), B, p->C, p->D, p->H, p->W, s));
    Buf coords, box, c_s2c;
    bool c2d_issued = false;
    auto finish_c2d = [&] {   // the per-frame box of voxels K3 will read (demand-driven final_conv)
        if (p->demand) {
            box = take(cs, (size_t)B * 8 * sizeof(int));
            RUN(cs, mphip_warp_sample_box(coords.p, (int *)box.p, B, p->D, p->H, p->W, cs.s));
        }
    };
    auto issue_c2d = [&] {
        if (c2d_issued) return;
        c2d_issued = true;
        if (!fork_wait()) { cs.rc = MPHIP_ELAUNCH; return; }
        GenLane one[1] = {{&cs, &p->c2d, Rd, td, zd, es}};
        Buf out1[1];
        generator_coords<1>(one, B, out1);
        coords = out1[0];
        finish_c2d();
    };
    if (p->demand && overlap && order == 1) {
        // the caller-stream chain first — its launches reach their hardware queue ahead of time, nothing in front of them waits —
        // THEN the side stream's wait and chain, which the runtime feeds in from the moment the step starts
        GenLane one[1] = {{&cm, &p->s2c, Rs, ts, zs, es}};
        Buf out1[1];
        generator_coords<1>(one, B, out1);
        c_s2c = out1[0];
        issue_c2d();
    } else if (p->demand && overlap) {
        if (!fork_wait()) return MPHIP_ELAUNCH;
        GenLane two[2] = {{&cm, &p->s2c, Rs, ts, zs, es}, {&cs, &p->c2d, Rd, td, zd, es}};
        Buf out2[2];
        generator_coords<2>(two, B, out2);
        c_s2c = out2[0];
        coords = out2[1];
        c2d_issued = true;
        finish_c2d();
    } else {
        GenLane one[1] = {{&cm, &p->s2c, Rs, ts, zs, es}};
        Buf out1[1];
        generator_coords<1>(one, B, out1);
        c_s2c = out1[0];
    }
    T5 vc = new_t5(cm, B, p->C, p->D, p->H, p->W, true);
    {
        const size_t wsb = mphip_warp_workspace_bytes(B, p->D, p->H, p->W);   // (covers the per-tile marks of the gather passes)
        Buf ws = take(cm, wsb);
        RUN(cm, mphip_warp_volume_coords_img(vs, c_s2c.p, vc.data.p, vc.range.p, B, p->C, p->D, p->H, p->W, ws.p, wsb, cimg.p, s));
        give(cm, ws);
    }
    give(cm, c_s2c);
    give(cm, cimg);
    int join_rc = MPHIP_OK;
    auto tail = [&]() -> Roi {   // the boxes are needed from here on: join the side stream before G3d's last upsample
        if (overlap && (hipEventRecord(p->ev_join, p->side) != hipSuccess || hipStreamWaitEvent(s, p->ev_join, 0) != hipSuccess)) {
            set_error("hot_slice_forward: stream join failed: %s", hipGetErrorString(hipGetLastError()));
            join_rc = MPHIP_ELAUNCH;
        }
        Roi r;
        r.on = p->demand != 0;
        r.box = (const int *)box.p;
        return r;
    };
    T5 vc2d = g3d(cm, vc, false, nullptr, issue_c2d, tail);
    if (cs.rc != MPHIP_OK && cm.rc == MPHIP_OK) cm.rc = cs.rc;
    if (join_rc != MPHIP_OK) return join_rc;
    // apply_warping_field + torch.sum(dim=2) (model.py:1167-1171) in one kernel (K3), on the coordinates computed above
    RUN(cm, mphip_warp_volume_dsum_coords(vc2d.data.p, coords.p, out, B, p->C, p->D, p->H, p->W, 0, s));
    give(cs, coords);
    give(cs, box);
    give(cm, vc2d);
    if (need) *need = p->main_arena.peak + p->side_arena.peak;
    return cm.rc;
}

int run_g3d(Plan *p, const float *x, const float *x_range, bool have_range, float *y, int B, void *workspace, size_t workspace_bytes, hipStream_t s,
            bool dry) {
    if (!dry) {
        run_g3d(p, nullptr, nullptr, have_range, nullptr, B, nullptr, 0, nullptr, true);
        if (workspace_bytes < p->main_arena.peak || !workspace) {
            set_error("g3d_forward: workspace %zu bytes < required %zu", workspace_bytes, p->main_arena.peak);
            return MPHIP_EWORKSPACE;
        }
    }
    const size_t cap = dry ? 0 : p->main_arena.peak;
    p->main_arena.reset(workspace, cap, dry);
    Ctx cm{p, &p->main_arena, s, dry};
    T5 xt;
    xt.n = B; xt.c = p->C; xt.d = p->D; xt.h = p->H; xt.w = p->W;
    xt.data.p = const_cast<float *>(x);
    xt.range.p = const_cast<float *>(x_range);
    xt.has_range = have_range;   // else range_for() measures x (one extra pass) into an arena descriptor
    // the caller's tensors are not arena blocks: give() ignores them (off == SIZE_MAX)
    T5 out = g3d(cm, xt, true, y, [] {}, [] { return Roi(); });
    (void)out;
    return cm.rc;
}

const float *lookup(const std::unordered_map<std::string, const float *> &sd, const std::string &key, std::string &missing) {
    auto it = sd.find(key);
    if (it == sd.end() || it->second == nullptr) {
        if (missing.size() < 300) missing += (missing.empty() ? "" : ", ") + key;
        return nullptr;
    }
    return it->second;
}

void bind_conv(ConvW &cw, const std::unordered_map<std::string, const float *> &sd, const std::string &prefix, int co, int ci, int k, std::string &missing) {
    cw.w = lookup(sd, prefix + ".weight", missing);
    cw.b = lookup(sd, prefix + ".bias", missing);
    cw.co = co; cw.ci = ci; cw.k = k;
    cw.fresh[0] = cw.fresh[1] = false;
}

void bind_generator(Generator &g, const std::unordered_map<std::string, const float *> &sd, const std::string &pre, int invert, std::string &missing) {
    static const int CH[5] = {512, 256, 128, 64, 32};
    g.invert = invert;
    g.gamma = lookup(sd, pre + "adaptive_matrix_gamma", missing);
    FlowFieldW &ff = g.ff;
    ff.w1x1 = lookup(sd, pre + "flowfield.conv1x1.weight", missing);
    ff.b1x1 = lookup(sd, pre + "flowfield.conv1x1.bias", missing);
    ff.kn_fresh = false;
    for (int i = 0; i < 4; ++i) {
        const std::string b = pre + "flowfield.resblock" + std::to_string(i + 1) + ".";
        ResBlockAda &rb = ff.rb[i];
        bind_conv(rb.conv1, sd, b + "conv1", CH[i + 1], CH[i], 3, missing);
        bind_conv(rb.conv2, sd, b + "conv2", CH[i + 1], CH[i + 1], 3, missing);
        bind_conv(rb.res, sd, b + "residual_conv", CH[i + 1], CH[i], 1, missing);
        rb.identity = false;
        Norm *ns[2] = {&rb.n1, &rb.n2};
        for (int j = 0; j < 2; ++j) {
            const std::string nb = b + "norm" + std::to_string(j + 1) + ".";
            ns[j]->gw = lookup(sd, nb + "group_norm.weight", missing);
            ns[j]->gb = lookup(sd, nb + "group_norm.bias", missing);
            ns[j]->w2 = lookup(sd, nb + "weight", missing);
            ns[j]->b2 = lookup(sd, nb + "bias", missing);
        }
    }
    bind_conv(ff.conv_out, sd, pre + "flowfield.conv3x3x3", 3, 32, 3, missing);
    ff.gn.gw = lookup(sd, pre + "flowfield.gn.weight", missing);
    ff.gn.gb = lookup(sd, pre + "flowfield.gn.bias", missing);
}

void bind_resblock(ResBlock &b, const std::unordered_map<std::string, const float *> &sd, const std::string &pre, int ci, int co, std::string &missing) {
    bind_conv(b.conv1, sd, pre + "conv1", co, ci, 3, missing);
    bind_conv(b.conv2, sd, pre + "conv2", co, co, 3, missing);
    b.identity = ci == co;
    if (!b.identity) bind_conv(b.shortcut, sd, pre + "shortcut", co, ci, 1, missing);
    b.gn1.gw = lookup(sd, pre + "gn1.weight", missing);
    b.gn1.gb = lookup(sd, pre + "gn1.bias", missing);
    b.gn2.gw = lookup(sd, pre + "gn2.weight", missing);
    b.gn2.gb = lookup(sd, pre + "gn2.bias", missing);
}

int bind_all(Plan *p, const char *const *names, const void *const *tensors, int n, bool generators) {
    std::unordered_map<std::string, const float *> sd;
    for (int i = 0; i < n; ++i)
        if (names[i]) sd[names[i]] = (const float *)tensors[i];
    std::string missing;
    if (generators) {
        bind_generator(p->s2c, sd, "warp_generator_s2c.", 1, missing);   // rigid part inverted (model.py:965)
        bind_generator(p->c2d, sd, "warp_generator_c2d.", 0, missing);
    }
    static const int DCH[5] = {96, 96, 192, 384, 768};
    for (int i = 0; i < 4; ++i) bind_resblock(p->down[i], sd, "G3d.downsampling." + std::to_string(2 * i) + ".", i == 0 ? p->C : DCH[i], DCH[i + 1], missing);
    static const int UCH[4] = {768, 384, 192, 96};
    for (int i = 0; i < 3; ++i) bind_resblock(p->up[i], sd, "G3d.upsampling." + std::to_string(2 * i) + ".", UCH[i], UCH[i + 1], missing);
    bind_conv(p->final_conv, sd, "G3d.final_conv", 96, 96, 3, missing);
    if (!missing.empty()) {
        set_error("hot_slice_plan: state-dict tensors missing: %s", missing.c_str());
        return MPHIP_EINVAL;
    }
    return MPHIP_OK;
}

int upload_table(Plan *p, int slot, float **dst, const uint32_t *bits, int n) {
    void *q = nullptr;
    if (hipMalloc(&q, (size_t)n * 4) != hipSuccess || hipMemcpy(q, bits, (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess) {
        set_error("hot_slice_plan: table upload failed");
        return MPHIP_ELAUNCH;
    }
    p->owned.push_back(q);
    *dst = (float *)q;
    p->own_lin[slot] = true;
    return MPHIP_OK;
}

const uint32_t *captured_linspace(int n) { return n == 16 ? MPHIP_TBL_LINSPACE_16 : n == 64 ? MPHIP_TBL_LINSPACE_64 : nullptr; }

}  // namespace

extern "C" int mphip_hot_slice_plan_create(const char *const *names, const void *const *tensors, int n_tensors, int C, int D, int H, int W,
                                           int flags, mphip_hot_slice_plan **out) {
    MPHIP_REQUIRE(names && tensors && out && n_tensors > 0, "hot_slice_plan_create: null pointer");
    MPHIP_REQUIRE(C == 96, "hot_slice_plan_create: the appearance volume has 96 channels (model.py:1157), got %d", C);
    MPHIP_REQUIRE(D > 0 && H > 0 && W > 0 && D % 8 == 0 && H % 8 == 0 && W % 8 == 0,
                  "hot_slice_plan_create: volume dims must be multiples of 8 (three 2x poolings), got %dx%dx%d", D, H, W);
    Plan *p = new Plan();
    p->C = C; p->D = D; p->H = H; p->W = W;
    p->have_generators = !(flags & MPHIP_PLAN_G3D_ONLY);
    p->overlap = !(flags & MPHIP_PLAN_SINGLE_STREAM);
    p->demand = !(flags & MPHIP_PLAN_FULL_FINAL_CONV);
    int rc = bind_all(p, names, tensors, n_tensors, p->have_generators);
    if (rc == MPHIP_OK && p->have_generators) {
        for (Generator *g : {&p->s2c, &p->c2d}) {
            void *q = nullptr;
            if (hipMalloc(&q, (size_t)512 * 2048 * 4) != hipSuccess) { set_error("hot_slice_plan_create: hipMalloc failed"); rc = MPHIP_ELAUNCH; break; }
            p->owned.push_back(q);
            g->ff.w1x1_kn = (float *)q;
            if (hipMalloc(&q, (size_t)512 * 2048 * 4) != hipSuccess) { set_error("hot_slice_plan_create: hipMalloc failed"); rc = MPHIP_ELAUNCH; break; }
            p->owned.push_back(q);
            g->ff.w_head = (float *)q;
        }
    }
    if (rc == MPHIP_OK) {   // built-in tables for the captured sizes; other sizes: mphip_hot_slice_plan_set_tables
        if (captured_linspace(D)) rc = upload_table(p, 0, &p->lin_d, captured_linspace(D), D);
        if (rc == MPHIP_OK && captured_linspace(H)) rc = upload_table(p, 1, &p->lin_h, captured_linspace(H), H);
        if (rc == MPHIP_OK && captured_linspace(W)) rc = upload_table(p, 2, &p->lin_w, captured_linspace(W), W);
        if (rc == MPHIP_OK) rc = upload_table(p, 3, &p->aff_base, MPHIP_TBL_AFFINE_BASE_64, 64);
    }
    if (rc == MPHIP_OK && p->overlap) {
        if (hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming) != hipSuccess) {
            set_error("hot_slice_plan_create: cannot create the side stream / events");
            rc = MPHIP_ELAUNCH;
        }
    }
    if (rc != MPHIP_OK) {
        mphip_hot_slice_plan_destroy(p);
        return rc;
    }
    *out = p;
    return MPHIP_OK;
}

extern "C" int mphip_hot_slice_plan_set_tables(mphip_hot_slice_plan *p, const float *lin_d, const float *lin_h, const float *lin_w,
                                               const float *affine_base) {
    MPHIP_REQUIRE(p, "hot_slice_plan_set_tables: null plan");
    if (lin_d) p->lin_d = const_cast<float *>(lin_d);
    if (lin_h) p->lin_h = const_cast<float *>(lin_h);
    if (lin_w) p->lin_w = const_cast<float *>(lin_w);
    if (affine_base) p->aff_base = const_cast<float *>(affine_base);
    return MPHIP_OK;
}

extern "C" int mphip_hot_slice_plan_set_precision(mphip_hot_slice_plan *p, int precision) {
    MPHIP_REQUIRE(p, "hot_slice_plan_set_precision: null plan");
    MPHIP_REQUIRE(precision == 0 || precision == 1, "hot_slice_plan_set_precision: 0 (exact fp32) or 1 (auto: f16x3 where supported), got %d", precision);
    if (p->precision != precision) {
        p->precision = precision;
        p->slice_sizes.clear();   // the workspace carve-up depends on the kernels chosen
    }
    return MPHIP_OK;
}

extern "C" int mphip_hot_slice_plan_profile(mphip_hot_slice_plan *p, int enable) {
    MPHIP_REQUIRE(p, "hot_slice_plan_profile: null plan");
    p->profile = enable != 0;
    p->prof_used[0] = p->prof_used[1] = 0;
    return MPHIP_OK;
}

// Sum and count of the launch durations recorded since the last read (synchronises on the recorded events): kind 0 = full
// launches of the dominant conv, 1 = its demand-driven launch.
extern "C" int mphip_hot_slice_plan_profile_read(mphip_hot_slice_plan *p, int kind, double *sum_ms, int *count) {
    MPHIP_REQUIRE(p && sum_ms && count && (kind == 0 || kind == 1), "hot_slice_plan_profile_read: bad arguments");
    double tot = 0.0;
    int cnt = 0;
    for (size_t i = 0; i + 1 < p->prof_used[kind]; i += 2) {
        float ms = 0.0f;
        if (hipEventSynchronize(p->prof_ev[kind][i + 1]) != hipSuccess || hipEventElapsedTime(&ms, p->prof_ev[kind][i], p->prof_ev[kind][i + 1]) != hipSuccess) {
            set_error("hot_slice_plan_profile_read: %s", hipGetErrorString(hipGetLastError()));
            return MPHIP_ELAUNCH;
        }
        tot += ms;
        ++cnt;
    }
    p->prof_used[kind] = 0;
    *sum_ms = tot;
    *count = cnt;
    return MPHIP_OK;
}

extern "C" int mphip_hot_slice_plan_refresh(mphip_hot_slice_plan *p, const char *const *names, const void *const *tensors, int n_tensors) {
    MPHIP_REQUIRE(p, "hot_slice_plan_refresh: null plan");
    if (names && tensors && n_tensors > 0) {   // new parameter storage (e.g. after .to()): re-bind, keep the pack buffers
        Plan tmp;
        tmp.C = p->C;
        int rc = bind_all(&tmp, names, tensors, n_tensors, p->have_generators);
        if (rc) return rc;
        auto rebind = [](ConvW &dst, const ConvW &src) { dst.w = src.w; dst.b = src.b; };
        auto rebind_gen = [&](Generator &d, const Generator &s) {
            d.gamma = s.gamma;
            d.ff.w1x1 = s.ff.w1x1; d.ff.b1x1 = s.ff.b1x1;
            for (int i = 0; i < 4; ++i) {
                rebind(d.ff.rb[i].conv1, s.ff.rb[i].conv1); rebind(d.ff.rb[i].conv2, s.ff.rb[i].conv2); rebind(d.ff.rb[i].res, s.ff.rb[i].res);
                d.ff.rb[i].n1 = s.ff.rb[i].n1; d.ff.rb[i].n2 = s.ff.rb[i].n2;
            }
            rebind(d.ff.conv_out, s.ff.conv_out);
            d.ff.gn = s.ff.gn;
        };
        if (p->have_generators) { rebind_gen(p->s2c, tmp.s2c); rebind_gen(p->c2d, tmp.c2d); }
        auto rebind_rb = [&](ResBlock &d, const ResBlock &s) {
            rebind(d.conv1, s.conv1); rebind(d.conv2, s.conv2);
            if (!d.identity) rebind(d.shortcut, s.shortcut);
            d.gn1 = s.gn1; d.gn2 = s.gn2;
        };
        for (int i = 0; i < 4; ++i) rebind_rb(p->down[i], tmp.down[i]);
        for (int i = 0; i < 3; ++i) rebind_rb(p->up[i], tmp.up[i]);
        rebind(p->final_conv, tmp.final_conv);
    }
    auto stale = [](ConvW &c) { c.fresh[0] = c.fresh[1] = false; };
    for (Generator *g : {&p->s2c, &p->c2d}) {
        g->ff.kn_fresh = false;
        for (int i = 0; i < 4; ++i) { stale(g->ff.rb[i].conv1); stale(g->ff.rb[i].conv2); stale(g->ff.rb[i].res); }
        stale(g->ff.conv_out);
    }
    for (int i = 0; i < 4; ++i) { stale(p->down[i].conv1); stale(p->down[i].conv2); stale(p->down[i].shortcut); }
    for (int i = 0; i < 3; ++i) { stale(p->up[i].conv1); stale(p->up[i].conv2); stale(p->up[i].shortcut); }
    stale(p->final_conv);
    return MPHIP_OK;
}

extern "C" size_t mphip_hot_slice_workspace_bytes(mphip_hot_slice_plan *p, int B) {
    if (!p || B <= 0 || !p->have_generators) return 0;
    const int chunk = B > MPHIP_PLAN_MAX_FRAMES_PER_PASS ? MPHIP_PLAN_MAX_FRAMES_PER_PASS : B;
    size_t need = 0;
    run_slice(p, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, chunk, nullptr, 0, nullptr, true, &need);
    return need;
}

extern "C" int mphip_hot_slice_forward(mphip_hot_slice_plan *p, const float *vs, const float *es, const float *Rs, const float *ts,
                                       const float *zs, const float *Rd, const float *td, const float *zd, float *out, int B, void *workspace,
                                       size_t workspace_bytes, void *stream) {
    MPHIP_REQUIRE(p && p->have_generators, "hot_slice_forward: null plan (or a G3d-only plan)");
    MPHIP_REQUIRE(vs && es && Rs && ts && zs && Rd && td && zd && out, "hot_slice_forward: null pointer");
    MPHIP_REQUIRE(B > 0, "hot_slice_forward: B=%d", B);
    MPHIP_REQUIRE(p->lin_d && p->lin_h && p->lin_w && p->aff_base,
                  "hot_slice_forward: no torch.linspace table for a %dx%dx%d volume is built in (16 and 64 are): call mphip_hot_slice_plan_set_tables",
                  p->D, p->H, p->W);
    const size_t vol = (size_t)p->C * p->D * p->H * p->W, plane = (size_t)p->C * p->H * p->W;
    // (the conv kernels address their input through one 2 GiB buffer resource: larger batches run as consecutive passes)
    for (int b0 = 0; b0 < B; b0 += MPHIP_PLAN_MAX_FRAMES_PER_PASS) {
        const int nb = B - b0 < MPHIP_PLAN_MAX_FRAMES_PER_PASS ? B - b0 : MPHIP_PLAN_MAX_FRAMES_PER_PASS;
        int rc = run_slice(p, vs + b0 * vol, es + (size_t)b0 * 512, Rs + b0 * 3, ts + b0 * 3, zs + (size_t)b0 * 512, Rd + b0 * 3, td + b0 * 3,
                           zd + (size_t)b0 * 512, out + b0 * plane, nb, workspace, workspace_bytes, (hipStream_t)stream, false, nullptr);
        if (rc) return rc;
    }
    return MPHIP_OK;
}

extern "C" size_t mphip_g3d_workspace_bytes(mphip_hot_slice_plan *p, int B) {
    if (!p || B <= 0) return 0;
    run_g3d(p, nullptr, nullptr, false, nullptr, B, nullptr, 0, nullptr, true);   // (sized for x_range == NULL: the larger case)
    return p->main_arena.peak;
}

extern "C" int mphip_g3d_forward(mphip_hot_slice_plan *p, const float *x, const float *x_range, float *y, int B, void *workspace,
                                 size_t workspace_bytes, void *stream) {
    MPHIP_REQUIRE(p && x && y, "g3d_forward: null pointer");
    MPHIP_REQUIRE(B > 0 && B <= MPHIP_PLAN_MAX_FRAMES_PER_PASS, "g3d_forward: 1 <= B <= %d frames per call, got %d", MPHIP_PLAN_MAX_FRAMES_PER_PASS, B);
    return run_g3d(p, x, x_range, x_range != nullptr, y, B, workspace, workspace_bytes, (hipStream_t)stream, false);
}

extern "C" void mphip_hot_slice_plan_destroy(mphip_hot_slice_plan *p) {
    if (!p) return;
    if (p->side) (void)hipStreamSynchronize(p->side);
    for (void *q : p->owned) (void)hipFree(q);
    for (int k = 0; k < 2; ++k)
        for (hipEvent_t e : p->prof_ev[k]) (void)hipEventDestroy(e);
    if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
    if (p->ev_join) (void)hipEventDestroy(p->ev_join);
    if (p->side) (void)hipStreamDestroy(p->side);
    delete p;
}
