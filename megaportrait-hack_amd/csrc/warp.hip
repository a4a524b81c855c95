// K1/K2/K3 — warp-field composition and the two volumetric warps of the Gbase hot slice.
// HBM-bound kernels: K2 is write-dominated (25 MB out per frame), K3 read-dominated.
// Reference call sites: model.py:965-973/1016-1022 (K1), model.py:1028-1065 (K2),
// model.py:1167-1171 (K3).  Built with -ffp-contract=off: every rounding below is placed where
// ATen's CPU kernels round; the FMAs ATen uses are explicit fmaf().
#include "mphip_ablate.h"
#include "mphip_common.h"
#include "mphip_resample.h"

namespace mphip {

// ----------------------------------------------------------------------------------------- K1
// One thread per (b,d,h,w); the three components share the index math.
__global__ void __launch_bounds__(256)
warp_field_compose_kernel(const float *__restrict__ theta, const float *__restrict__ em,
                          const float *__restrict__ base, float *__restrict__ wout, float *__restrict__ rt_out,
                          float *__restrict__ em_out, int B, int eD, int eH, int eW, int G) {
    MPHIP_LATENCY_KERNEL_PRIO();
    const size_t vol = (size_t)G * G * G;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)B * vol) return;
    int b = (int)(t / vol);
    size_t r = t - (size_t)b * vol;
    int d = (int)(r / ((size_t)G * G));
    int h = (int)((r / G) % G);
    int w = (int)(r % G);
    const float x = base[w], y = base[h], z = base[d];
    const SrcIdx sd = src_index<false>(d, eD, G), sh = src_index<false>(h, eH, G), sw = src_index<false>(w, eW, G);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float *th = theta + ((size_t)b * 3 + j) * 4;
        float acc = x * th[0];
        acc = fmaf(y, th[1], acc);
        acc = fmaf(z, th[2], acc);
        acc = fmaf(1.0f, th[3], acc);
        float e = trilerp(em + ((size_t)b * 3 + j) * eD * eH * eW, eH, eW, sd, sh, sw);
        size_t o = ((size_t)b * 3 + j) * vol + r;
        wout[o] = acc + e;
        if (rt_out) rt_out[o] = acc;
        if (em_out) em_out[o] = e;
    }
}

// ------------------------------------------------------------------------------------ K2 / K3
// Sample coordinate of output voxel (b,d,h,w): the literal op chain of model.py:1036-1058 and
// ATen GridSampler.h:27-36,58-60.  Returns the clipped un-normalised coordinate per axis.
struct Coord3 {
    float x, y, z;
};

__device__ __forceinline__ float coord_axis(float g, float f, float sz) {
    float p = g + f;         // model.py:1052  grid + warp_field
    float m = 2.0f * p;      // model.py:1058  2.0 * warped_grid
    float q = m / sz;        //                / normalization_factors
    float n = q - 1.0f;      //                - 1.0
    float c = ((n + 1.0f) / 2.0f) * sz;  // grid_sampler_unnormalize, align_corners=True
    return fminf(sz, fmaxf(c, 0.0f));    // clip_coordinates (padding_mode='border')
}

__device__ __forceinline__ Coord3 sample_coord(const float *__restrict__ field, const float *__restrict__ lin_d,
                                               const float *__restrict__ lin_h, const float *__restrict__ lin_w,
                                               int b, int d, int h, int w, int D, int H, int W, int fD, int fH,
                                               int fW) {
    const SrcIdx sd = src_index<true>(d, fD, D), sh = src_index<true>(h, fH, H), sw = src_index<true>(w, fW, W);
    const size_t fvol = (size_t)fD * fH * fW;
    const float *fb = field + (size_t)b * 3 * fvol;
    Coord3 c;
    c.x = coord_axis(lin_w[w], trilerp(fb, fH, fW, sd, sh, sw), (float)(W - 1));
    c.y = coord_axis(lin_h[h], trilerp(fb + fvol, fH, fW, sd, sh, sw), (float)(H - 1));
    c.z = coord_axis(lin_d[d], trilerp(fb + 2 * fvol, fH, fW, sd, sh, sw), (float)(D - 1));
    return c;
}

// K1 + the coordinate pass in one kernel (r03): the warp field [B,3,G,G,G] a generator composes is read exactly once, by the
// coordinate pass of the warp it feeds, which only needs the 2 x D of its G depth planes the align_corners=True resize touches.
// One thread per (b,d,h,w) of the (D,G,G) volume evaluates warp_field_compose_kernel's expression at the two source planes and
// warp_coords_kernel<INPLANE>'s chain on them — the same operations in the same order: bit-identical coordinates, without the
// 2 x 25 MB round trip and one launch less on the latency-bound chain.
__device__ __forceinline__ float compose_value(const float *__restrict__ th /* theta[b][j] */, const float *__restrict__ emj, float x, float y,
                                               float z, int eH, int eW, const SrcIdx &sd, const SrcIdx &sh, const SrcIdx &sw) {
    float acc = x * th[0];
    acc = fmaf(y, th[1], acc);
    acc = fmaf(z, th[2], acc);
    acc = fmaf(1.0f, th[3], acc);
    return acc + trilerp(emj, eH, eW, sd, sh, sw);
}

__global__ void __launch_bounds__(256)
warp_field_coords_kernel(const float *__restrict__ theta, const float *__restrict__ em, const float *__restrict__ base,
                         const float *__restrict__ lin_d, const float *__restrict__ lin_h, const float *__restrict__ lin_w,
                         float *__restrict__ coords, int B, int eD, int eH, int eW, int G, int D) {
    MPHIP_LATENCY_KERNEL_PRIO();
    const size_t n = (size_t)B * D * G * G;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int w = (int)(t % G);
    size_t r = t / G;
    const int h = (int)(r % G);
    r /= G;
    const int d = (int)(r % D);
    const int b = (int)(r / D);
    const SrcIdx sd = src_index<true>(d, G, D);   // the field's depth planes this output slice blends (warp_coords_kernel<true>)
    const float x = base[w], y = base[h];
    const SrcIdx eh = src_index<false>(h, eH, G), ew = src_index<false>(w, eW, G);
    const SrcIdx e0 = src_index<false>(sd.i0, eD, G), e1 = src_index<false>(sd.i1, eD, G);
    const float z0 = base[sd.i0], z1 = base[sd.i1];
    float f[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float *th = theta + ((size_t)b * 3 + j) * 4;
        const float *emj = em + ((size_t)b * 3 + j) * eD * eH * eW;
        const float f0 = compose_value(th, emj, x, y, z0, eH, eW, e0, eh, ew);
        const float f1 = compose_value(th, emj, x, y, z1, eH, eW, e1, eh, ew);
        f[j] = lerp2(sd.l0, f0, sd.l1, f1);
    }
    coords[t * 3] = coord_axis(lin_w[w], f[0], (float)(G - 1));
    coords[t * 3 + 1] = coord_axis(lin_h[h], f[1], (float)(G - 1));
    coords[t * 3 + 2] = coord_axis(lin_d[d], f[2], (float)(D - 1));
}

// 8-tap trilinear gather set-up for one voxel: base offset of the (z0,y0,x0) corner, the deltas
// to the +1 corners (0 when that corner is outside: ATen skips it, its weight is 0 there), and
// the 8 corner weights in ATen's accumulation order tnw,tne,tsw,tse,bnw,bne,bsw,bse.
struct Taps {
    int base, dx, dy, dz;
    float w[8];
};

__device__ __forceinline__ Taps make_taps(const Coord3 &c, int D, int H, int W) {
    Taps t;
    int x0 = (int)floorf(c.x), y0 = (int)floorf(c.y), z0 = (int)floorf(c.z);
    float wx1 = c.x - (float)x0, wx0 = (float)(x0 + 1) - c.x;
    float wy1 = c.y - (float)y0, wy0 = (float)(y0 + 1) - c.y;
    float wz1 = c.z - (float)z0, wz0 = (float)(z0 + 1) - c.z;
    bool vx = x0 + 1 < W, vy = y0 + 1 < H, vz = z0 + 1 < D;
    t.base = (z0 * H + y0) * W + x0;
    t.dx = vx ? 1 : 0;
    t.dy = vy ? W : 0;
    t.dz = vz ? H * W : 0;
    if (!vx) wx1 = 0.0f;
    if (!vy) wy1 = 0.0f;
    if (!vz) wz1 = 0.0f;
    t.w[0] = wx0 * wy0 * wz0;
    t.w[1] = wx1 * wy0 * wz0;
    t.w[2] = wx0 * wy1 * wz0;
    t.w[3] = wx1 * wy1 * wz0;
    t.w[4] = wx0 * wy0 * wz1;
    t.w[5] = wx1 * wy0 * wz1;
    t.w[6] = wx0 * wy1 * wz1;
    t.w[7] = wx1 * wy1 * wz1;
    return t;
}

__device__ __forceinline__ float gather8(const float *__restrict__ vol, const Taps &t) {
    const float *p = vol + t.base;
    float acc = 0.0f;
    acc += p[0] * t.w[0];
    acc += p[t.dx] * t.w[1];
    acc += p[t.dy] * t.w[2];
    acc += p[t.dy + t.dx] * t.w[3];
    acc += p[t.dz] * t.w[4];
    acc += p[t.dz + t.dx] * t.w[5];
    acc += p[t.dz + t.dy] * t.w[6];
    acc += p[t.dz + t.dy + t.dx] * t.w[7];
    return acc;
}

// The same 8 taps with the two x-neighbours of every (y,z) corner fetched by ONE 8-byte load (4 loads instead of 8; only
// dword alignment is needed).  Same values, same accumulation order -> bit-identical to gather8.  At the right border
// (dx == 0: the +x corner is outside, ATen skips it, its weight is 0) the pair is read one voxel to the left and both taps
// take its second element, i.e. p[0] — exactly what gather8 reads there.  Needs W >= 2.
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ float gather8_pairs(const float *__restrict__ vol, const Taps &t) {
    const float *p = vol + t.base - (t.dx ? 0 : 1);
    const f32x2u q0 = *reinterpret_cast<const f32x2u *>(p);
    const f32x2u q1 = *reinterpret_cast<const f32x2u *>(p + t.dy);
    const f32x2u q2 = *reinterpret_cast<const f32x2u *>(p + t.dz);
    const f32x2u q3 = *reinterpret_cast<const f32x2u *>(p + t.dz + t.dy);
    const bool in = t.dx != 0;
    float acc = 0.0f;
    acc += (in ? q0.x : q0.y) * t.w[0];
    acc += q0.y * t.w[1];
    acc += (in ? q1.x : q1.y) * t.w[2];
    acc += q1.y * t.w[3];
    acc += (in ? q2.x : q2.y) * t.w[4];
    acc += q2.y * t.w[5];
    acc += (in ? q3.x : q3.y) * t.w[6];
    acc += q3.y * t.w[7];
    return acc;
}

// ---- coordinate pass -------------------------------------------------------------------------
// One thread per output voxel: coords[B,D,H,W,3] = clipped (x,y,z) sample coordinates (and the
// floor indices for the tests).  12 B per voxel (0.79 MB per 512^2 frame, 3 % of K2's traffic);
// K2/K3 read it back instead of re-deriving the chain per channel slice.
// INPLANE: fH==H && fW==W -> the align_corners=True resize is the identity in H,W (weights exactly
// (1,0)), so only the depth lerp remains — bitwise the same value as the full 8-corner form.
template <bool INPLANE>
__global__ void __launch_bounds__(256)
warp_coords_kernel(const float *__restrict__ field, const float *__restrict__ lin_d, const float *__restrict__ lin_h,
                   const float *__restrict__ lin_w, float *__restrict__ coords, int32_t *__restrict__ idx, int B,
                   int D, int H, int W, int fD, int fH, int fW) {
    const size_t n = (size_t)B * D * H * W;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    int w = (int)(t % W);
    size_t r = t / W;
    int h = (int)(r % H);
    r /= H;
    int d = (int)(r % D);
    int b = (int)(r / D);
    Coord3 c;
    if (INPLANE) {
        const SrcIdx sd = src_index<true>(d, fD, D);
        const size_t fvol = (size_t)fD * fH * fW;
        const float *f0 = field + (size_t)b * 3 * fvol + ((size_t)sd.i0 * fH + h) * fW + w;
        const float *f1 = field + (size_t)b * 3 * fvol + ((size_t)sd.i1 * fH + h) * fW + w;
        c.x = coord_axis(lin_w[w], lerp2(sd.l0, f0[0], sd.l1, f1[0]), (float)(W - 1));
        c.y = coord_axis(lin_h[h], lerp2(sd.l0, f0[fvol], sd.l1, f1[fvol]), (float)(H - 1));
        c.z = coord_axis(lin_d[d], lerp2(sd.l0, f0[2 * fvol], sd.l1, f1[2 * fvol]), (float)(D - 1));
    } else {
        c = sample_coord(field, lin_d, lin_h, lin_w, b, d, h, w, D, H, W, fD, fH, fW);
    }
    coords[t * 3] = c.x;
    coords[t * 3 + 1] = c.y;
    coords[t * 3 + 2] = c.z;
    if (idx) {
        idx[t * 3] = (int)floorf(c.x);
        idx[t * 3 + 1] = (int)floorf(c.y);
        idx[t * 3 + 2] = (int)floorf(c.z);
    }
}

// ---- gather pass -----------------------------------------------------------------------------
// The source voxels a tile of output voxels needs form a small box when the warp is smooth (the
// reference's own fields move samples by a few voxels: SURVEY.md §0 quirk 1).  Each workgroup
// finds that bounding box with a wavefront-shuffle + LDS min/max reduction, stages the box for a
// slice of channels into LDS with coalesced row reads, and does the 8-tap trilinear gather from
// LDS (neighbouring lanes hit the same or adjacent words: broadcast, no bank conflicts).  A box
// that does not fit (wild fields) falls back to gathering from global memory.
constexpr int STAGE_FLOATS = 12288;  // 48 KB of LDS for the staged box

struct Box {
    int ox, oy, oz, ex, ey, ez;
};

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = min(v, __shfl_xor(v, s, 64));
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v = max(v, __shfl_xor(v, s, 64));
    return v;
}

// Block-wide bounding box of the (x0,y0,z0) corners, extended by the +1 corner and clamped (NW waves; red: NW * 6 ints of LDS).
template <int NW>
__device__ __forceinline__ Box block_box_n(int lx, int ly, int lz, int hx, int hy, int hz, int D, int H, int W, int *red) {
    lx = wave_min(lx); ly = wave_min(ly); lz = wave_min(lz);
    hx = wave_max(hx); hy = wave_max(hy); hz = wave_max(hz);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[wave * 6 + 0] = lx; red[wave * 6 + 1] = ly; red[wave * 6 + 2] = lz;
        red[wave * 6 + 3] = hx; red[wave * 6 + 4] = hy; red[wave * 6 + 5] = hz;
    }
    __syncthreads();
    int m[6] = {red[0], red[1], red[2], red[3], red[4], red[5]};
#pragma unroll
    for (int w = 1; w < NW; ++w) {
        m[0] = min(m[0], red[w * 6]); m[1] = min(m[1], red[w * 6 + 1]); m[2] = min(m[2], red[w * 6 + 2]);
        m[3] = max(m[3], red[w * 6 + 3]); m[4] = max(m[4], red[w * 6 + 4]); m[5] = max(m[5], red[w * 6 + 5]);
    }
    Box bx;
    bx.ox = m[0]; bx.oy = m[1]; bx.oz = m[2];
    bx.ex = min(m[3] + 1, W - 1) - bx.ox + 1;
    bx.ey = min(m[4] + 1, H - 1) - bx.oy + 1;
    bx.ez = min(m[5] + 1, D - 1) - bx.oz + 1;
    return bx;
}
__device__ __forceinline__ Box block_box(int lx, int ly, int lz, int hx, int hy, int hz, int D, int H, int W,
                                         int *red /* >= 24 ints of LDS */) {
    lx = wave_min(lx); ly = wave_min(ly); lz = wave_min(lz);
    hx = wave_max(hx); hy = wave_max(hy); hz = wave_max(hz);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[wave * 6 + 0] = lx; red[wave * 6 + 1] = ly; red[wave * 6 + 2] = lz;
        red[wave * 6 + 3] = hx; red[wave * 6 + 4] = hy; red[wave * 6 + 5] = hz;
    }
    __syncthreads();
    Box bx;
    bx.ox = min(min(red[0], red[6]), min(red[12], red[18]));
    bx.oy = min(min(red[1], red[7]), min(red[13], red[19]));
    bx.oz = min(min(red[2], red[8]), min(red[14], red[20]));
    int mx = max(max(red[3], red[9]), max(red[15], red[21]));
    int my = max(max(red[4], red[10]), max(red[16], red[22]));
    int mz = max(max(red[5], red[11]), max(red[17], red[23]));
    bx.ex = min(mx + 1, W - 1) - bx.ox + 1;
    bx.ey = min(my + 1, H - 1) - bx.oy + 1;
    bx.ez = min(mz + 1, D - 1) - bx.oz + 1;
    return bx;
}

// Stage channels [c0, c0+cs) of the box into lds[z][y][x][c] with an odd channel pitch cs_pad: a tap's LDS
// address is then the same for every channel up to an immediate offset (no per-channel address arithmetic in
// the gather loop) and lanes that read different voxels hit different banks.  Lane -> box element (decoded
// once per 64-element chunk), waves stride over channels.
__device__ __forceinline__ void stage_box(const float *__restrict__ vb /* v + b*C*vol */, float *lds, const Box &bx,
                                          int c0, int cs, int cs_pad, int H, int W, size_t vol) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int exy = bx.ex * bx.ey, bvol = exy * bx.ez;
    for (int r = lane; r < bvol; r += 64) {
        int z = r / exy, r2 = r - z * exy;
        int y = r2 / bx.ex, x = r2 - y * bx.ex;
        const float *src = vb + (size_t)c0 * vol + ((size_t)(bx.oz + z) * H + bx.oy + y) * W + bx.ox + x;
        float *dst = lds + r * cs_pad;
        // eight channel planes per trip, loads first: a load -> LDS-store trip at a time costs one L2 round trip per trip (24 of them
        // for 96 channels; r03: that was half of K2's 57 us on the reference's fields)
        for (int c = wave; c < cs; c += 32) {
            float t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = c + 4 * k < cs ? src[(size_t)(c + 4 * k) * vol] : 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (c + 4 * k < cs) dst[c + 4 * k] = t[k];
        }
    }
}

// Planar image lds[c][z][y][x] (K2: scalar taps, one pass over <= 12288/bvol channels; staging and gathers are
// conflict-free because lanes walk consecutive box elements).
__device__ __forceinline__ void stage_box_planar(const float *__restrict__ vb, float *lds, const Box &bx, int c0, int cs,
                                                 int H, int W, size_t vol) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int exy = bx.ex * bx.ey, bvol = exy * bx.ez;
    for (int r = lane; r < bvol; r += 64) {
        int z = r / exy, r2 = r - z * exy;
        int y = r2 / bx.ex, x = r2 - y * bx.ex;
        const float *src = vb + (size_t)c0 * vol + ((size_t)(bx.oz + z) * H + bx.oy + y) * W + bx.ox + x;
        float *dst = lds + r;
        for (int c = wave; c < cs; c += 32) {   // (eight planes per trip, loads first: see stage_box)
            float t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = c + 4 * k < cs ? src[(size_t)(c + 4 * k) * vol] : 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (c + 4 * k < cs) dst[(c + 4 * k) * bvol] = t[k];
        }
    }
}

// tap offsets of a voxel re-expressed in the staged LDS image (the 8 weights stay in the Taps)
struct TapOff {
    int base, dx, dy, dz;
};

__device__ __forceinline__ TapOff rebase(const Taps &t, int x0, int y0, int z0, const Box &bx, int cs_pad) {
    TapOff r;
    r.base = (((z0 - bx.oz) * bx.ey + (y0 - bx.oy)) * bx.ex + (x0 - bx.ox)) * cs_pad;
    r.dx = t.dx ? cs_pad : 0;
    r.dy = t.dy ? bx.ex * cs_pad : 0;
    r.dz = t.dz ? bx.ex * bx.ey * cs_pad : 0;
    return r;
}

__device__ __forceinline__ float gather8_lds(const float *__restrict__ img, const TapOff &o, const float (&w)[8]) {
    const float *p = img + o.base;
    float acc = 0.0f;
    acc += p[0] * w[0];
    acc += p[o.dx] * w[1];
    acc += p[o.dy] * w[2];
    acc += p[o.dy + o.dx] * w[3];
    acc += p[o.dz] * w[4];
    acc += p[o.dz + o.dx] * w[5];
    acc += p[o.dz + o.dy] * w[6];
    acc += p[o.dz + o.dy + o.dx] * w[7];
    return acc;
}

// Four channels per tap with one ds_read_b128 each (LDS image [voxel][channel], 16-byte aligned pitch): every
// channel sees exactly the scalar op sequence of gather8 (acc=0; acc += p*w in tap order), so values stay
// bit-identical, at a quarter of the LDS instructions.
__device__ __forceinline__ void gather8x4(const float *__restrict__ img, const TapOff &t, const float (&tw)[8],
                                          float out[4]) {
    const float *p = img + t.base;
    const float4 q0 = *reinterpret_cast<const float4 *>(p);
    const float4 q1 = *reinterpret_cast<const float4 *>(p + t.dx);
    const float4 q2 = *reinterpret_cast<const float4 *>(p + t.dy);
    const float4 q3 = *reinterpret_cast<const float4 *>(p + t.dy + t.dx);
    const float4 q4 = *reinterpret_cast<const float4 *>(p + t.dz);
    const float4 q5 = *reinterpret_cast<const float4 *>(p + t.dz + t.dx);
    const float4 q6 = *reinterpret_cast<const float4 *>(p + t.dz + t.dy);
    const float4 q7 = *reinterpret_cast<const float4 *>(p + t.dz + t.dy + t.dx);
    // channel pairs on the packed fp32 pipe (v_pk_mul_f32 / v_pk_add_f32: per channel still acc = 0; acc += q_k * w_k in tap order,
    // one rounding per op — the same bits as the scalar sequence at half the instructions; K3 on the reference's fields 52.7 -> 48.4 us, r04)
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    const float4 q[8] = {q0, q1, q2, q3, q4, q5, q6, q7};
    f32x2_ lo = {0.0f, 0.0f}, hi = {0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const f32x2_ wk = {tw[k], tw[k]};
        lo += f32x2_{q[k].x, q[k].y} * wk;
        hi += f32x2_{q[k].z, q[k].w} * wk;
    }
    out[0] = lo[0]; out[1] = lo[1]; out[2] = hi[0]; out[3] = hi[1];
}

// channel pitch of the staged image: a multiple of 4 floats (16-byte tap reads) whose quarter is odd, so lanes
// that read different voxels fall on different 16-byte bank slots
__device__ __forceinline__ int lds_pitch_for(int channels) {
    int p = (channels + 3) & ~3;
    if (((p >> 2) & 1) == 0) p += 4;
    return p;
}

// K2, tiles whose samples all lie in the volume's LOW CORNER — every tile of the reference's own fields (apply_warping_field hands
// grid_sample coordinates of size ~[-2, 3] as if they were voxel indices).  Any other tile — a field that really
// travels through the volume — is only MARKED here (todo[tile]) and done by warp_gather_columns_kernel / warp_gather_direct_kernel
// below: keeping those paths out of this kernel keeps it lean.
//
// How it got its shape (r04; wall-clock stamps per workgroup, tools/dbg_k2_trace.py, and ablations, tools/k2_ablate.sh, B=8):
//   * r03 (32 x 32 tile, four positions per thread, every workgroup staging its own source box line by line from global memory —
//     1.5 k scattered 128-byte lines, the rows of all 96 planes on the same few L2 channels): 63 us.  Not the stores: without them
//     59 us.  Not the tap arithmetic either: half the VALU instructions (packed fp32), no address arithmetic in the loop, half the
//     dependent chain per wave at four waves per SIMD — each +-0.  The stamps: the gather LOOPS wrote at ~7.7 TB/s, the speed of a
//     plain fill of the same 201 MB, but covered less than half of a workgroup's life; in front of them sat three dependent
//     round trips (coordinates -> box -> staging loads), and a tail of workgroups whose staging loads queued behind the store flood
//     of the others ended at 60 us when the median had ended at 34.
//   * So the corner [0,E)^3 of every frame is copied ONCE per call (warp_corner_image_kernel; the hot slice does it at the start of
//     its step, vs is an input) into a compact image that already has the LDS layout, [frame][channel group][cell][channels + pad],
//     contiguous — all L2 channels serve it — and a workgroup brings its block in by LDS-DMA, issued FIRST THING: it lands under
//     the coordinate loads and the box reduction.  One global round trip in front of the stores instead of three.
//   * More, smaller workgroups (3072 of 256 threads, three resident rounds) gave 50 us: every workgroup pays the coordinate round
//     trip (3.7 us) before it stores for 7 us.  One workgroup per CU and launch does better: 1024 threads own a 32 x 64 tile of one
//     (b,d) plane, two positions per thread, ALL channels (image: 6^3 cells x 98 floats = 85 KB), one prologue, then nothing but
//     tap reads and stores.  Few tiles (B = 1, 2): the channels are split over blockIdx.y so that every CU has work.
//   * The image is [cell][channel], channel pitch = channels + 2 (98): a tap's (c, c+1) pair is ONE 8-byte-aligned ds_read_b64 whose
//     channel is an immediate offset (no address arithmetic in the loop), and lanes that read different cells fall on different
//     bank pairs.  What bounds the loop now is its arithmetic — 16 fp32 multiplies / adds per output value, kept as separate,
//     separately rounded ops for ATen's bits: 20 us of the chip's fp32 pipe at B=8, 34 us with addressing and stores; the packed ops
//     (v_pk_mul_f32 / v_pk_add_f32, half the instructions) measured the same time as scalar ones.
// Values: per channel and position exactly gather8's op sequence (acc = 0; acc += p_k * w_k in tap order, one rounding per op).
constexpr int K2_TH = 32, K2_TW = 64;      // 2048 positions per workgroup, two per thread
constexpr int K2_THREADS = 1024;
constexpr int K2_DIRECT_SPLIT = 4;         // channel groups of the direct gather (warp_gather_direct_body)
constexpr int K2_COLUMNS_MAX_BOX = 16384;  // source-box voxels of a tile up to which the column walk is used
constexpr int K2_CORNER_E = 6;
constexpr int K2_CORNER_CELLS = K2_CORNER_E * K2_CORNER_E * K2_CORNER_E;
constexpr int K2_CG_MAX = 96;              // channels per workgroup at most (LDS: 216 x 98 floats = 84672 B)
constexpr int K2_LDS_FLOATS = K2_CORNER_CELLS * (K2_CG_MAX + 2);
// channels per workgroup (blockIdx.y groups): all of them (<= 96) when the tiles alone fill the chip, else 32 or 16
__host__ __device__ inline int k2_group_channels(size_t tiles, int C) {
    int cg = min(C, K2_CG_MAX);
    if (tiles * (size_t)((C + cg - 1) / cg) < 256 && C > 32) cg = 32;
    if (tiles * (size_t)((C + cg - 1) / cg) < 256 && C > 16) cg = 16;
    return cg;
}
// channel pitch of the image: EVEN (a tap's channel pair is one 8-byte-aligned ds_read_b64: 256 B/clk, twice ds_read2_b32) and = 2 mod 32
// for the channel counts in use (98, 34, 18 -> cell * pitch mod 64 takes 32 different even values: the 32 lanes of a read group that
// hit different cells fall on different bank pairs)
__host__ __device__ inline int k2_pitch(int cg) { return ((cg + 1) & ~1) + 2; }
// image: [frame][group][cell][k2_pitch(cg)] floats, a group's block padded to a multiple of 16 bytes
__host__ __device__ inline size_t k2_block_floats(int cg) { return ((size_t)K2_CORNER_CELLS * k2_pitch(cg) + 3) / 4 * 4; }
__global__ void __launch_bounds__(128)
warp_corner_image_kernel(const float *__restrict__ v, float *__restrict__ img, int C, int D, int H, int W, int cg, int groups) {
    const int cell = blockIdx.x, b = blockIdx.y;
    const int z = cell / (K2_CORNER_E * K2_CORNER_E), y = (cell / K2_CORNER_E) % K2_CORNER_E, x = cell % K2_CORNER_E;
    const bool inside = z < D && y < H && x < W;
    const size_t vol = (size_t)D * H * W, blk = k2_block_floats(cg);
    const int cgp = k2_pitch(cg);
    for (int c = threadIdx.x; c < groups * cgp; c += 128) {   // (the pad slot and channels >= C: zeros, the block is copied whole)
        const int g = c / cgp, cl = c - g * cgp, ch = g * cg + cl;
        const bool real = inside && cl < cg && ch < C;
        img[((size_t)b * groups + g) * blk + (size_t)cell * cgp + cl] = real ? v[((size_t)b * C + ch) * vol + ((size_t)z * H + y) * W + x] : 0.0f;
    }
}
// one block of the corner image -> LDS, as LDS-DMA (16 bytes per lane, 1 KiB per wave and instruction, no registers); issued by hand:
// the compiler would wait for each transfer before the next LDS access.  The caller waits (vmcnt(0)) before its barrier.
__device__ __forceinline__ void k2_dma_image(const float *__restrict__ blk, float *lds, int floats) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)lds;
    const int pieces = (floats + 255) / 256;
    for (int q = wave; q < pieces; q += K2_THREADS / 64) {
        const int f = q * 256 + lane * 4;   // float index of this lane's 16 bytes
        if (f < floats) {
            const float *src = blk + f;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds0 + (unsigned)q * 1024u) : "memory");
        }
    }
}
// ... and without an image (a caller whose workspace has no room for it): the workgroup collects its block itself
__device__ __forceinline__ void k2_stage_corner(const float *__restrict__ vb, float *lds, int c0, int cs, int cgp, int D, int H, int W) {
    const size_t vol = (size_t)D * H * W;
    for (int i = threadIdx.x; i < K2_CORNER_CELLS * cs; i += K2_THREADS) {
        const int c = i / K2_CORNER_CELLS, cell = i - c * K2_CORNER_CELLS;   // lanes along cells: neighbours share lines
        const int z = cell / (K2_CORNER_E * K2_CORNER_E), y = (cell / K2_CORNER_E) % K2_CORNER_E, x = cell % K2_CORNER_E;
        lds[cell * cgp + c] = (z < D && y < H && x < W) ? vb[(size_t)(c0 + c) * vol + ((size_t)z * H + y) * W + x] : 0.0f;
    }
}

// The tap arithmetic on the PACKED fp32 pipe, two channels per instruction: the (c, c+1) pair of a tap is exactly what one ds_read2_b32
// returns, and the tap's weight is broadcast to both halves by op_sel — weights stay single registers (pairs {w_2j, w_2j+1}, even / odd
// picked by the instruction's op_sel bits).  Written as inline assembly because hipcc's own packing of this loop duplicates every weight
// into a register pair (+180 registers, 700 B of scratch at 128).  With the staging gone the loop is VALU-bound (95 -> 52 instructions per
// channel pair).  Per channel and position still acc = 0; acc = acc + p_k * w_k in tap order, one rounding per op.
typedef float k2_f2 __attribute__((ext_vector_type(2)));
template <int ODD>
__device__ __forceinline__ k2_f2 k2_pk_mul(k2_f2 p, k2_f2 wpair) {
    k2_f2 m;
    if (ODD) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(m) : "v"(p), "v"(wpair));
    else     asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(m) : "v"(p), "v"(wpair));
    return m;
}
__device__ __forceinline__ k2_f2 k2_pk_add(k2_f2 a, k2_f2 b) {
    k2_f2 m;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b));
    return m;
}
#ifdef MPHIP_K2_TRACE   /* dev: wall-clock (100 MHz) stamps per workgroup: start, box known, image staged, done */
__device__ unsigned long long g_k2_trace[4096 * 4];
#define K2_STAMP(i) if (threadIdx.x == 0 && blockIdx.y * gridDim.x + blockIdx.x < 4096) g_k2_trace[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + (i)] = wall_clock64();
#else
#define K2_STAMP(i)
#endif
__global__ void __launch_bounds__(K2_THREADS)
warp_gather_kernel(const float *__restrict__ v, const float *__restrict__ coords, float *__restrict__ out,
                   float *__restrict__ out_range /* optional range descriptor of `out`: G3d's first conv reads it */,
                   int *__restrict__ todo, int B, int C, int D, int H, int W,
                   const float *__restrict__ img /* optional corner image (warp_corner_image_kernel, same cg) */, int cg /* channels per blockIdx.y */) {
    __shared__ __attribute__((aligned(16))) float lds[K2_LDS_FLOATS];
    __shared__ int red[(K2_THREADS / 64) * 6];
    K2_STAMP(0)
    const int HW = H * W;
    const int tiles_w = (W + K2_TW - 1) / K2_TW, tiles_h = (H + K2_TH - 1) / K2_TH;
    // XCD-aware order: consecutive logical ids (d fastest, then tile, then frame) run on the same XCD
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int d = (int)(bid % (unsigned)D);
    const int tile = (int)((bid / (unsigned)D) % (unsigned)(tiles_w * tiles_h));
    const int b = (int)(bid / ((unsigned)D * (unsigned)(tiles_w * tiles_h)));
    const int cg0 = (int)blockIdx.y * cg, Cg = min(C - cg0, cg), cgp = k2_pitch(cg);
    const int h = (tile / tiles_w) * K2_TH + (int)(threadIdx.x / (K2_TW / 2));
    const int w = (tile % tiles_w) * K2_TW + (int)(threadIdx.x % (K2_TW / 2)) * 2;
    const bool active = h < H && w < W;  // W % 4 == 0 -> a thread's 2 positions share validity
    const int p0 = h * W + w;
    const size_t vol = (size_t)D * HW;

    Taps taps[2];
    int x0[2], y0[2], z0[2];
    int lx = INT_MAX, ly = INT_MAX, lz = INT_MAX, hx = 0, hy = 0, hz = 0;
    float cf[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (active) {   // (issued BEFORE the scalar test below is waited for: one round trip for both)
        const float *cp = coords + (((size_t)b * D + d) * HW + p0) * 3;   // (p0 even: 8-byte aligned)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float2 t2 = *reinterpret_cast<const float2 *>(cp + q * 2);
            cf[q * 2] = t2.x; cf[q * 2 + 1] = t2.y;
        }
    }
    // The image is fetched only if the tile's FIRST sample lies in the corner (one scalar load): a field that travels through the volume
    // (not the reference's) would otherwise pay for the transfers for nothing.
    const float *c0p = coords + (((size_t)b * D + d) * HW + (size_t)(tile / tiles_w) * K2_TH * W + (tile % tiles_w) * K2_TW) * 3;
    const float fx = c0p[0], fy = c0p[1], fz = c0p[2];
    const bool maybe = fx >= 0.0f && fx < (float)(K2_CORNER_E - 1) && fy >= 0.0f && fy < (float)(K2_CORNER_E - 1) && fz >= 0.0f && fz < (float)(K2_CORNER_E - 1);
    const bool dma = maybe && img != nullptr;
#ifndef MPHIP_K2_ABL_NOSTAGE   /* dev ablations (timing only, wrong results): tools/k2_ablate.sh */
    if (dma) k2_dma_image(img + ((size_t)b * gridDim.y + blockIdx.y) * k2_block_floats(cg), lds, (int)k2_block_floats(cg));
#endif
    if (active) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            Coord3 c{cf[i * 3], cf[i * 3 + 1], cf[i * 3 + 2]};
            taps[i] = make_taps(c, D, H, W);
            x0[i] = (int)floorf(c.x); y0[i] = (int)floorf(c.y); z0[i] = (int)floorf(c.z);
            lx = min(lx, x0[i]); ly = min(ly, y0[i]); lz = min(lz, z0[i]);
            hx = max(hx, x0[i]); hy = max(hy, y0[i]); hz = max(hz, z0[i]);
        }
    }
    const Box bx = block_box_n<K2_THREADS / 64>(lx, ly, lz, hx, hy, hz, D, H, W, red);
    K2_STAMP(1)
    // block-uniform: every sample of the tile (all eight corners of each) inside the corner the image holds
    const bool in_corner = bx.ox + bx.ex <= K2_CORNER_E && bx.oy + bx.ey <= K2_CORNER_E && bx.oz + bx.ez <= K2_CORNER_E;
    // 0: done here; 1: a box of moderate size = a smooth field that travels -> warp_gather_columns_body (plane reuse down the
    // slices); 2: no locality to exploit (a box like the whole volume) -> warp_gather_direct_body (most loads in flight)
    if (threadIdx.x == 0 && blockIdx.y == 0) todo[bid] = in_corner ? 0 : (bx.ex * bx.ey * bx.ez <= K2_COLUMNS_MAX_BOX ? 1 : 2);
    unsigned mbits = 0;
    if (dma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the image have landed
    if (in_corner) {
        float *ob = out + (size_t)b * C * vol + (size_t)d * HW + p0;
#ifndef MPHIP_K2_ABL_NOSTAGE
        if (!dma) k2_stage_corner(v + (size_t)b * C * vol, lds, cg0, Cg, cgp, D, H, W);
#endif
        __syncthreads();
        K2_STAMP(2)
        if (active) {
            int tb[2][8];   // tap addresses in the image (floats, premultiplied by the pitch)
            const Box cbx{0, 0, 0, K2_CORNER_E, K2_CORNER_E, K2_CORNER_E};
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const TapOff o = rebase(taps[i], x0[i], y0[i], z0[i], cbx, cgp);
                tb[i][0] = o.base; tb[i][1] = o.base + o.dx; tb[i][2] = o.base + o.dy; tb[i][3] = o.base + o.dy + o.dx;
                tb[i][4] = o.base + o.dz; tb[i][5] = o.base + o.dz + o.dx; tb[i][6] = o.base + o.dz + o.dy;
                tb[i][7] = o.base + o.dz + o.dy + o.dx;
            }
            k2_f2 wp[2][4];   // the taps' weights in pairs
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) wp[i][j] = k2_f2{taps[i].w[2 * j], taps[i].w[2 * j + 1]};
            auto two_channels = [&](const float *src, int c) {
                k2_f2 pv[2][8];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int k = 0; k < 8; ++k) pv[i][k] = *reinterpret_cast<const k2_f2 *>(src + tb[i][k]);   // (even pitch, even channel: 8-byte aligned)
                k2_f2 acc[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    acc[i] = k2_f2{0.0f, 0.0f};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[i] = k2_pk_add(acc[i], k2_pk_mul<0>(pv[i][2 * j], wp[i][j]));
                        acc[i] = k2_pk_add(acc[i], k2_pk_mul<1>(pv[i][2 * j + 1], wp[i][j]));
                    }
                }
#ifdef MPHIP_K2_ABL_NOSTORE
                if (acc[0][0] == 1.2345e30f)
#endif
                {
                    *reinterpret_cast<float2 *>(ob + (size_t)(cg0 + c) * vol) = make_float2(acc[0][0], acc[1][0]);
                    *reinterpret_cast<float2 *>(ob + (size_t)(cg0 + c + 1) * vol) = make_float2(acc[0][1], acc[1][1]);
                }
                mbits = max(max(mbits, range_bits(acc[0][0])), max(range_bits(acc[0][1]), max(range_bits(acc[1][0]), range_bits(acc[1][1]))));
                __builtin_amdgcn_sched_barrier(0);   // one channel pair at a time (hoisting the next pairs' reads spills)
            };
            int c = 0;
#ifndef MPHIP_K2_ABL_NOLOOP
            for (; c + 8 <= Cg; c += 8) {   // (eight channels per trip: their offsets are immediates of the tap reads)
                const float *src = lds + c;
#pragma unroll
                for (int u = 0; u < 8; u += 2) two_channels(src + u, c + u);
            }
            for (; c + 2 <= Cg; c += 2) two_channels(lds + c, c);
            if (c < Cg) {   // odd tail
                float r[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    r[i] = 0.0f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) r[i] += lds[c + tb[i][k]] * taps[i].w[k];
                }
                *reinterpret_cast<float2 *>(ob + (size_t)(cg0 + c) * vol) = make_float2(r[0], r[1]);
                mbits = max(mbits, max(range_bits(r[0]), range_bits(r[1])));
            }
#endif
        }
    }
    // (slot = group * tiles + tile: the follow-up kernels fold into group 0's slots)
    if (out_range) range_note_block(mbits, out_range, blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    K2_STAMP(3)
}

// The tiles warp_gather_kernel marked: one position per lane, lanes running along w, so for a smooth field every tap load of
// a wave covers one or two contiguous row segments (the per-CU L1 serves the overlap between taps and rows) and the stores
// are contiguous 128-byte rows; the x-neighbour taps come in pairs (gather8_pairs).  Workgroups of unmarked tiles exit.
__device__ __forceinline__ void
warp_gather_direct_body(const float *__restrict__ v, const float *__restrict__ coords, float *__restrict__ out,
                        float *__restrict__ out_range, const int *__restrict__ todo, int B, int C, int D, int H, int W,
                        unsigned blk_x, unsigned grid_x, unsigned blk_y, unsigned grid_y) {
    unsigned mbits = 0;
    const unsigned bid = xcd_remap(blk_x, grid_x);  // same logical order as warp_gather_kernel
    if (todo[bid] == 2) {  // block-uniform
        const int HW = H * W;
        const int tiles_w = (W + K2_TW - 1) / K2_TW, tiles_h = (H + K2_TH - 1) / K2_TH;
        const int d = (int)(bid % (unsigned)D);
        const int tile = (int)((bid / (unsigned)D) % (unsigned)(tiles_w * tiles_h));
        const int b = (int)(bid / ((unsigned)D * (unsigned)(tiles_w * tiles_h)));
        const size_t vol = (size_t)D * HW;
        // a workgroup takes four passes of 256 positions (RPP rows each): blk_y = (row part of the tile) * K2_DIRECT_SPLIT + channel group
        constexpr int RPP = 256 / K2_TW, NP = 4;
        static_assert(K2_TH % (NP * RPP) == 0, "tile rows");
        const int part = (int)blk_y / K2_DIRECT_SPLIT;
        blk_y %= K2_DIRECT_SPLIT; grid_y = K2_DIRECT_SPLIT;
        const int w = (tile % tiles_w) * K2_TW + (int)(threadIdx.x % K2_TW);
        const int hb = (tile / tiles_w) * K2_TH + part * NP * RPP + (int)(threadIdx.x / K2_TW);  // rows hb, hb+RPP, ...
        Taps t[NP];
        bool act[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int h = hb + RPP * i;
            act[i] = h < H && w < W;
            if (act[i]) {
                const float *cq = coords + (((size_t)b * D + d) * HW + h * W + w) * 3;
                t[i] = make_taps(Coord3{cq[0], cq[1], cq[2]}, D, H, W);
            }
        }
        const float *vb = v + (size_t)b * C * vol;
        float *ob = out + (size_t)b * C * vol + (size_t)d * HW + hb * W + w;
        // channels are split over grid_y workgroups: this path is latency-bound (L2-hit gathers), it needs every wave slot
        const int cpg = (C + (int)grid_y - 1) / (int)grid_y;
        const int c_end = min(C, ((int)blk_y + 1) * cpg);
#pragma unroll 2
        for (int c = (int)blk_y * cpg; c < c_end; ++c) {
            const float *src = vb + (size_t)c * vol;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                if (act[i]) {
                    const float r = W >= 2 ? gather8_pairs(src, t[i]) : gather8(src, t[i]);
                    ob[(size_t)c * vol + RPP * i * W] = r;
                    mbits = max(mbits, range_bits(r));
                }
            }
        }
    }
    // the rare path folds its maximum into the slot warp_gather_kernel wrote for this tile (one atomic per wave that did work;
    // idle workgroups leave without touching the descriptor)
    if (out_range && todo[bid] == 2) {
        mbits = wave_umax(mbits);
        if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned *>(out_range) + 4 + bid, mbits);
    }
}

// The same marked tiles, walked like K3 walks them: a workgroup owns a 16 x 16 tile of (h,w) positions, ALL D output slices and
// K2C_CPB channels; every thread runs down the slices of its position (taps once per slice, channels innermost).  Consecutive
// output slices of a smooth field sample neighbouring source planes, so the plane a slice fetched is still in the L1 / L2 when
// the next slice needs it — the direct gather, one workgroup per (tile, slice), re-fetches it from another CU.
#ifndef MPHIP_K2C_CPB
#define MPHIP_K2C_CPB 16
#endif
constexpr int K2C_CPB = MPHIP_K2C_CPB;
__device__ __forceinline__ void
warp_gather_columns_body(const float *__restrict__ v, const float *__restrict__ coords, float *__restrict__ out,
                         float *__restrict__ out_range, const int *__restrict__ todo, int B, int C, int D, int H, int W,
                         unsigned blk, unsigned nblk) {
    const int HW = H * W;
    const int tiles_w = (W + 15) / 16, tiles_h = (H + 15) / 16, ntile = tiles_w * tiles_h;
    const int tiles32_w = (W + K2_TW - 1) / K2_TW, ntile32 = tiles32_w * ((H + K2_TH - 1) / K2_TH);
    unsigned bid = xcd_remap(blk, nblk);  // the tiles of one (frame, channel slice) under one L2 (as K3)
    const int tile = (int)(bid % ntile); bid /= ntile;
    const int slices = (C + K2C_CPB - 1) / K2C_CPB;
    const int slice = (int)(bid % slices), b = (int)(bid / slices);
    const int h = (tile / tiles_w) * 16 + (int)(threadIdx.x >> 4);
    const int w = (tile % tiles_w) * 16 + (int)(threadIdx.x & 15);
    const bool active = h < H && w < W;
    const int p = h * W + w;
    const size_t vol = (size_t)D * HW;
    const int c0 = slice * K2C_CPB, cs = min(K2C_CPB, C - c0);
    // the 16 x 16 tile lies inside one 32 x 32 tile of warp_gather_kernel: its marks, one per slice (block-uniform)
    const int t32 = ((tile / tiles_w) * 16 / K2_TH) * tiles32_w + (tile % tiles_w) * 16 / K2_TW;
    const int *marks = todo + ((size_t)b * ntile32 + t32) * D;
    const float *vb = v + ((size_t)b * C + c0) * vol;
    float *ob = out + ((size_t)b * C + c0) * vol + p;
    unsigned mbits = 0;
    // nothing marked for this column (every launch on the reference's own fields): leave after ONE round of loads
    int mine = 0;
    for (int d = threadIdx.x; d < D; d += 256) mine |= marks[d] == 1;
    if (!__syncthreads_or(mine)) return;
    for (int d = 0; d < D; ++d) {
        if (marks[d] != 1 || !active) continue;
        const float *q = coords + (((size_t)b * D + d) * HW + p) * 3;
        const Taps t = make_taps(Coord3{q[0], q[1], q[2]}, D, H, W);
#pragma unroll
        for (int c = 0; c < K2C_CPB; ++c)
            if (c < cs) {
                const float r = gather8_pairs(vb + (size_t)c * vol, t);
                ob[(size_t)c * vol + (size_t)d * HW] = r;
                mbits = max(mbits, range_bits(r));
            }
    }
    if (out_range) {  // into the slot warp_gather_kernel wrote for (frame, 32 x 32 tile, slice 0)
        mbits = wave_umax(mbits);
        if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned *>(out_range) + 4 + ((size_t)b * ntile32 + t32) * D, mbits);
    }
}

// Two launches, not one kernel with two roles: merged, the column walk inherits the direct gather's 223 registers (two waves
// per SIMD) and loses what it gained (smooth field 223 -> 325 us); the second, mostly idle launch costs the reference-field path ~4 us.
__global__ void __launch_bounds__(256)
warp_gather_columns_kernel(const float *__restrict__ v, const float *__restrict__ coords, float *__restrict__ out,
                           float *__restrict__ out_range, const int *__restrict__ todo, int B, int C, int D, int H, int W) {
    warp_gather_columns_body(v, coords, out, out_range, todo, B, C, D, H, W, blockIdx.x, gridDim.x);
}
__global__ void __launch_bounds__(256)
warp_gather_direct_kernel(const float *__restrict__ v, const float *__restrict__ coords, float *__restrict__ out,
                          float *__restrict__ out_range, const int *__restrict__ todo, int B, int C, int D, int H, int W) {
    warp_gather_direct_body(v, coords, out, out_range, todo, B, C, D, H, W, blockIdx.x, gridDim.x, blockIdx.y, gridDim.y);
}

// K3: a workgroup owns a compact 16 x 16 tile of (h,w) positions of one frame and CPB channels; every thread walks the D
// output slices of its position accumulating the depth projection in registers (d ascending, like torch.sum(dim=2) on
// the warped volume, which is never written).  If the source box of ALL D slices fits the LDS image (the reference's own
// fields: a 4^3 corner) it is staged once, [voxel][channel] with 16-byte tap reads; otherwise the taps are gathered from
// global memory through the L1 (lanes along w, x-neighbours in pairs).
#ifndef MPHIP_K3_TH
#define MPHIP_K3_TH 16
#endif
constexpr int K3_TH = MPHIP_K3_TH, K3_TW = 256 / MPHIP_K3_TH;
template <int CPB>
__global__ void __launch_bounds__(256)
warp_gather_dsum_kernel(const float *__restrict__ v, const float *__restrict__ coords, float *__restrict__ out,
                        int B, int C, int D, int H, int W, size_t v_frame_stride /* floats; 0 = one shared source volume */) {
    __shared__ __attribute__((aligned(16))) float lds[STAGE_FLOATS];
    __shared__ int red[24];
    const int HW = H * W;
    const int tiles_w = (W + K3_TW - 1) / K3_TW, tiles_h = (H + K3_TH - 1) / K3_TH, ntile = tiles_w * tiles_h;
    // XCD-aware order: the tiles of one (frame, channel slice) are consecutive logical ids, i.e. they run on ONE XCD at about the
    // same time — w-neighbours share every 128-byte line of a source row, h-neighbours the halo rows, and with the hardware's
    // round-robin (tile t -> XCD t % 8) each of those lines was fetched into up to four different L2s (travelling fields: 5x the
    // algorithmic bytes crossed the fabric, at 6.9 TB/s — the kernel's limit)
    unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = (int)(bid % ntile); bid /= ntile;
    const int slices = (C + CPB - 1) / CPB;
    const int slice = (int)(bid % slices), b = (int)(bid / slices);
    static_assert(K3_TH * K3_TW == 256, "one thread per position of the tile");
    const int h = (tile / tiles_w) * K3_TH + (int)(threadIdx.x / K3_TW);
    const int w = (tile % tiles_w) * K3_TW + (int)(threadIdx.x % K3_TW);
    const bool active = h < H && w < W;
    const int p = h * W + w;
    const size_t vol = (size_t)D * HW;
    const int c0 = slice * CPB;
    const int cs = min(CPB, C - c0);
    const int cs_pad = lds_pitch_for(cs);
    const float *cp = coords + ((size_t)b * D * HW + (active ? p : 0)) * 3;
    const float *vb = v + (size_t)b * v_frame_stride;

    float acc[CPB];
#pragma unroll
    for (int c = 0; c < CPB; ++c) acc[c] = 0.0f;

    int lx = INT_MAX, ly = INT_MAX, lz = INT_MAX, hx = 0, hy = 0, hz = 0;
    if (active) {
        for (int d = 0; d < D; ++d) {
            const float *q = cp + (size_t)d * HW * 3;
            int x = (int)floorf(q[0]), y = (int)floorf(q[1]), z = (int)floorf(q[2]);
            lx = min(lx, x); ly = min(ly, y); lz = min(lz, z);
            hx = max(hx, x); hy = max(hy, y); hz = max(hz, z);
        }
    }
    const Box all = block_box(lx, ly, lz, hx, hy, hz, D, H, W, red);
    if (all.ex * all.ey * all.ez * cs_pad <= STAGE_FLOATS) {  // block-uniform: everything in one [voxel][channel] image
        stage_box(vb, lds, all, c0, cs, cs_pad, H, W, vol);
        __syncthreads();
        if (active) {
            for (int d = 0; d < D; ++d) {
                const float *q = cp + (size_t)d * HW * 3;
                Coord3 cc{q[0], q[1], q[2]};
                Taps t = make_taps(cc, D, H, W);
                const TapOff lt = rebase(t, (int)floorf(cc.x), (int)floorf(cc.y), (int)floorf(cc.z), all, cs_pad);
#pragma unroll
                for (int c = 0; c < CPB; c += 4) {
                    if (c + 4 <= cs) {
                        float tmp[4];
                        gather8x4(lds + c, lt, t.w, tmp);
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[c + k] += tmp[k];
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (c + k < cs) acc[c + k] += gather8_lds(lds + c + k, lt, t.w);
                    }
                }
            }
        }
    } else if (active) {
        // a field that travels through the volume: gather from global memory; lanes run along w (coalesced row segments for
        // a smooth field, the per-CU L1 serves the overlap between taps), x-neighbour taps in pairs
        for (int d = 0; d < D; ++d) {
            const float *q = cp + (size_t)d * HW * 3;
            const Taps t = make_taps(Coord3{q[0], q[1], q[2]}, D, H, W);
#pragma unroll
            for (int c = 0; c < CPB; ++c)
                if (c < cs) acc[c] += W >= 2 ? gather8_pairs(vb + (size_t)(c0 + c) * vol, t) : gather8(vb + (size_t)(c0 + c) * vol, t);
        }
    }
    if (!active) return;
#pragma unroll
    for (int c = 0; c < CPB; ++c)
        if (c < cs) out[((size_t)b * C + c0 + c) * HW + p] = acc[c];
}

// Fallback for W % 4 != 0 (never the case on the hot path): one thread per output voxel and channel slice.
__global__ void __launch_bounds__(256)
warp_gather_scalar_kernel(const float *__restrict__ v, const float *__restrict__ coords, float *__restrict__ out,
                          int B, int C, int D, int H, int W, int cpb) {
    const size_t vol = (size_t)D * H * W;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)B * vol) return;
    int b = (int)(t / vol);
    size_t r = t - (size_t)b * vol;
    Coord3 c{coords[t * 3], coords[t * 3 + 1], coords[t * 3 + 2]};
    Taps taps = make_taps(c, D, H, W);
    const int c_begin = blockIdx.y * cpb, c_end = min(C, c_begin + cpb);
    for (int ch = c_begin; ch < c_end; ++ch)
        out[((size_t)b * C + ch) * vol + r] = gather8(v + ((size_t)b * C + ch) * vol, taps);
}

}  // namespace mphip

using namespace mphip;

extern "C" int mphip_warp_field_compose(const float *theta, const float *em, const float *base_tbl, float *w,
                                        float *rt_out, float *em_out, int B, int eD, int eH, int eW, int G,
                                        void *stream) {
    MPHIP_REQUIRE(theta && em && base_tbl && w, "warp_field_compose: null pointer");
    MPHIP_REQUIRE(B > 0 && eD > 0 && eH > 0 && eW > 0 && G > 0, "warp_field_compose: bad dims");
    size_t n = (size_t)B * G * G * G;
    hipLaunchKernelGGL(warp_field_compose_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, theta, em,
                       base_tbl, w, rt_out, em_out, B, eD, eH, eW, G);
    return check_launch("warp_field_compose");
}

static int check_warp_args(const char *name, const void *v, const void *field, const void *ld, const void *lh,
                           const void *lw, const void *out, int B, int C, int D, int H, int W, int fD, int fH,
                           int fW) {
    MPHIP_REQUIRE(v && field && ld && lh && lw && out, "%s: null pointer", name);
    MPHIP_REQUIRE(B > 0 && C > 0 && D > 0 && H > 0 && W > 0 && fD > 0 && fH > 0 && fW > 0, "%s: bad dims", name);
    MPHIP_REQUIRE((size_t)D * H * W < (1u << 30), "%s: volume too large for 32-bit tap offsets", name);
    return MPHIP_OK;
}

static size_t k2_tiles(int B, int D, int H, int W) {
    return (size_t)B * D * ((H + K2_TH - 1) / K2_TH) * ((W + K2_TW - 1) / K2_TW);
}

static size_t k2_todo_bytes(int B, int D, int H, int W) { return ((k2_tiles(B, D, H, W) * sizeof(int) + 15) / 16) * 16; }
// coordinates [B,D,H,W,3] + one int per K2 tile (which of the two gather kernels takes it)
extern "C" size_t mphip_warp_workspace_bytes(int B, int D, int H, int W) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)B * D * H * W * 3 * sizeof(float) + k2_todo_bytes(B, D, H, W);
}
// K2's optional corner image (warp_corner_image_kernel): a workspace that is this much larger than the entry point's minimum lets the
// gather stage low-corner boxes — every box of the reference's own fields — from one compact copy per frame
extern "C" size_t mphip_warp_corner_image_bytes(int B, int C) {
    if (B <= 0 || C <= 0) return 0;
    size_t fl = 0;   // [frame][channel group][6^3 cells][k2_pitch(cg)], for whichever grouping the launch picks
    for (int cg : {min(C, K2_CG_MAX), 32, 16})
        if (cg <= C) fl = std::max(fl, (size_t)cdiv(C, cg) * k2_block_floats(cg));
    return (size_t)B * fl * sizeof(float);
}

static int launch_coords(const float *field, const float *lin_d, const float *lin_h, const float *lin_w, float *coords,
                         int32_t *idx, int B, int D, int H, int W, int fD, int fH, int fW, hipStream_t s) {
    size_t n = (size_t)B * D * H * W;
    if (fH == H && fW == W)
        hipLaunchKernelGGL(warp_coords_kernel<true>, dim3(cdiv(n, 256)), dim3(256), 0, s, field, lin_d, lin_h, lin_w,
                           coords, idx, B, D, H, W, fD, fH, fW);
    else
        hipLaunchKernelGGL(warp_coords_kernel<false>, dim3(cdiv(n, 256)), dim3(256), 0, s, field, lin_d, lin_h, lin_w,
                           coords, idx, B, D, H, W, fD, fH, fW);
    return check_launch("warp_coords");
}

static int warp_volume_gather(const float *v, const float *coords, float *out, float *out_range, int *todo, float *corner_img, bool img_ready,
                              int B, int C, int D, int H, int W, hipStream_t s);

extern "C" int mphip_warp_volume(const float *v, const float *field, const float *lin_d, const float *lin_h,
                                 const float *lin_w, float *out, float *coords_out, int32_t *idx_out, float *out_range, int B,
                                 int C, int D, int H, int W, int fD, int fH, int fW, void *workspace, size_t workspace_bytes,
                                 void *stream) {
    int rc = check_warp_args("warp_volume", v, field, lin_d, lin_h, lin_w, out, B, C, D, H, W, fD, fH, fW);
    if (rc) return rc;
    MPHIP_REQUIRE(!idx_out || coords_out, "warp_volume: idx_out requires coords_out");
    const size_t need = mphip_warp_workspace_bytes(B, D, H, W), coord_bytes = (size_t)B * D * H * W * 3 * sizeof(float);
    if (!workspace || workspace_bytes < need) {
        set_error("warp_volume: workspace %zu bytes < required %zu", workspace_bytes, need);
        return MPHIP_EWORKSPACE;
    }
    float *coords = coords_out ? coords_out : (float *)workspace;
    int *todo = (int *)((char *)workspace + coord_bytes);
    float *corner_img = workspace_bytes >= need + mphip_warp_corner_image_bytes(B, C) ? (float *)((char *)workspace + need) : nullptr;
    hipStream_t s = (hipStream_t)stream;
    rc = launch_coords(field, lin_d, lin_h, lin_w, coords, idx_out, B, D, H, W, fD, fH, fW, s);
    if (rc) return rc;
    return warp_volume_gather(v, coords, out, out_range, todo, corner_img, false, B, C, D, H, W, s);
}

// the gather pass(es) of K2 on given coordinates; todo: one int per tile; corner_img: optional mphip_warp_corner_image_bytes(B, C) bytes
static int warp_volume_gather(const float *v, const float *coords, float *out, float *out_range, int *todo, float *corner_img, bool img_ready,
                              int B, int C, int D, int H, int W, hipStream_t s) {
    int rc;
    const size_t nblocks = k2_tiles(B, D, H, W);
    // channel groups of the corner gather (its range slots: one per workgroup)
    const int cg = k2_group_channels(nblocks, C);
    const unsigned groups = (unsigned)cdiv(C, cg);
    if (out_range && (W % 4 != 0 || nblocks * groups > RANGE_MAX_PARTS)) {
        // (scalar fallback kernel / more workgroups than partial slots) the warp is a convex combination of v's voxels:
        // max|out| <= max|v|, so v's own range serves
        rc = absmax_range_launch(v, (size_t)B * C * D * H * W, out_range, s);
        if (rc) return rc;
        out_range = nullptr;
    }
    if (W % 4 == 0) {
        const unsigned ncol = (unsigned)((size_t)B * ((H + 15) / 16) * ((W + 15) / 16) * cdiv(C, K2C_CPB));
        if (corner_img && !img_ready)
            hipLaunchKernelGGL(warp_corner_image_kernel, dim3(K2_CORNER_CELLS, (unsigned)B), dim3(128), 0, s, v, corner_img, C, D, H, W, cg, (int)groups);
        hipLaunchKernelGGL(warp_gather_kernel, dim3((unsigned)nblocks, groups), dim3(K2_THREADS), 0, s, v, coords, out, out_range, todo, B, C, D, H, W,
                           (const float *)corner_img, cg);
        // the tiles it marked: smooth travelling fields -> column walk, incoherent ones -> direct gather (workgroups of the other
        // kind, and all of them on the reference's own fields, exit after one load)
        hipLaunchKernelGGL(warp_gather_columns_kernel, dim3(ncol), dim3(256), 0, s, v, (const float *)coords, out, out_range,
                           (const int *)todo, B, C, D, H, W);
        hipLaunchKernelGGL(warp_gather_direct_kernel, dim3((unsigned)nblocks, K2_DIRECT_SPLIT * (K2_TH / (4 * (256 / K2_TW)))), dim3(256), 0, s, v, (const float *)coords,
                           out, out_range, (const int *)todo, B, C, D, H, W);
    } else {
        const int cpb = C >= 48 ? 12 : C;
        hipLaunchKernelGGL(warp_gather_scalar_kernel, dim3(cdiv((size_t)B * D * H * W, 256), cdiv(C, cpb)), dim3(256), 0, s,
                           v, coords, out, B, C, D, H, W, cpb);
    }
    return check_launch("warp_volume");
}

static int warp_volume_dsum_impl(const char *name, const float *v, size_t v_frame_stride, const float *field,
                                 const float *lin_d, const float *lin_h, const float *lin_w, float *out, int B, int C, int D,
                                 int H, int W, int fD, int fH, int fW, void *workspace, size_t workspace_bytes, void *stream) {
    int rc = check_warp_args(name, v, field, lin_d, lin_h, lin_w, out, B, C, D, H, W, fD, fH, fW);
    if (rc) return rc;
    size_t need = mphip_warp_workspace_bytes(B, D, H, W);
    if (!workspace || workspace_bytes < need) {
        set_error("%s: workspace %zu bytes < required %zu", name, workspace_bytes, need);
        return MPHIP_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    float *coords = (float *)workspace;
    rc = launch_coords(field, lin_d, lin_h, lin_w, coords, nullptr, B, D, H, W, fD, fH, fW, s);
    if (rc) return rc;
    constexpr int CPB = 16;
    const int tiles = ((H + K3_TH - 1) / K3_TH) * ((W + K3_TW - 1) / K3_TW);
    hipLaunchKernelGGL(warp_gather_dsum_kernel<CPB>, dim3((unsigned)((size_t)B * tiles * cdiv(C, CPB))), dim3(256), 0, s, v,
                       coords, out, B, C, D, H, W, v_frame_stride);
    return check_launch(name);
}

// K1 + the coordinate pass fused (warp_field_coords_kernel): theta [B,3,4], em [B,3,eD,eH,eW] -> coords [B,D,G,G,3] of the warp of a
// (D,G,G) volume by the composed field — bit-identical to mphip_warp_field_compose + mphip_warp_coords, the field never exists.
extern "C" int mphip_warp_field_coords(const float *theta, const float *em, const float *base_tbl, const float *lin_d, const float *lin_h,
                                       const float *lin_w, float *coords, int B, int eD, int eH, int eW, int G, int D, void *stream) {
    MPHIP_REQUIRE(theta && em && base_tbl && lin_d && lin_h && lin_w && coords, "warp_field_coords: null pointer");
    MPHIP_REQUIRE(B > 0 && eD > 0 && eH > 0 && eW > 0 && G > 0 && D > 0, "warp_field_coords: bad dims");
    const size_t n = (size_t)B * D * G * G;
    hipLaunchKernelGGL(warp_field_coords_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, theta, em, base_tbl, lin_d, lin_h, lin_w,
                       coords, B, eD, eH, eW, G, D);
    return check_launch("warp_field_coords");
}

// K2 on given coordinates (mphip_warp_coords / mphip_warp_field_coords); workspace: one int per 32x32 tile
// (mphip_warp_workspace_bytes covers it).
extern "C" int mphip_warp_volume_coords(const float *v, const float *coords, float *out, float *out_range, int B, int C, int D, int H, int W,
                                        void *workspace, size_t workspace_bytes, void *stream) {
    MPHIP_REQUIRE(v && coords && out, "warp_volume_coords: null pointer");
    MPHIP_REQUIRE(B > 0 && C > 0 && D > 0 && H > 0 && W > 0, "warp_volume_coords: bad dims");
    MPHIP_REQUIRE((size_t)D * H * W < (1u << 30), "warp_volume_coords: volume too large for 32-bit tap offsets");
    const size_t need = k2_todo_bytes(B, D, H, W);
    if (!workspace || workspace_bytes < need) {
        set_error("warp_volume_coords: workspace %zu bytes < required %zu", workspace_bytes, need);
        return MPHIP_EWORKSPACE;
    }
    float *corner_img = workspace_bytes >= need + mphip_warp_corner_image_bytes(B, C) ? (float *)((char *)workspace + need) : nullptr;
    return warp_volume_gather(v, coords, out, out_range, (int *)workspace, corner_img, false, B, C, D, H, W, (hipStream_t)stream);
}

// The corner image on its own: a caller that has `v` long before the coordinates (the hot slice: vs is an INPUT, the coordinates come out
// of an 18-launch generator chain) builds it early, off the critical path, and hands it to mphip_warp_volume_coords_img.
extern "C" int mphip_warp_corner_image(const float *v, void *img, size_t img_bytes, int B, int C, int D, int H, int W, void *stream) {
    MPHIP_REQUIRE(v && img, "warp_corner_image: null pointer");
    MPHIP_REQUIRE(B > 0 && C > 0 && D > 0 && H > 0 && W > 0, "warp_corner_image: bad dims");
    MPHIP_REQUIRE(((uintptr_t)img & 15) == 0, "warp_corner_image: img must be 16-byte aligned");
    if (img_bytes < mphip_warp_corner_image_bytes(B, C)) {
        set_error("warp_corner_image: buffer %zu bytes < required %zu", img_bytes, mphip_warp_corner_image_bytes(B, C));
        return MPHIP_EWORKSPACE;
    }
    const int cg = k2_group_channels(k2_tiles(B, D, H, W), C);
    hipLaunchKernelGGL(warp_corner_image_kernel, dim3(K2_CORNER_CELLS, (unsigned)B), dim3(128), 0, (hipStream_t)stream, v, (float *)img, C, D, H, W,
                       cg, cdiv(C, cg));
    return check_launch("warp_corner_image");
}
extern "C" int mphip_warp_volume_coords_img(const float *v, const float *coords, float *out, float *out_range, int B, int C, int D, int H, int W,
                                            void *workspace, size_t workspace_bytes, const void *img, void *stream) {
    MPHIP_REQUIRE(v && coords && out && img, "warp_volume_coords_img: null pointer");
    MPHIP_REQUIRE(B > 0 && C > 0 && D > 0 && H > 0 && W > 0, "warp_volume_coords_img: bad dims");
    MPHIP_REQUIRE((size_t)D * H * W < (1u << 30), "warp_volume_coords_img: volume too large for 32-bit tap offsets");
    MPHIP_REQUIRE(((uintptr_t)img & 15) == 0, "warp_volume_coords_img: img must be 16-byte aligned");
    const size_t need = k2_todo_bytes(B, D, H, W);
    if (!workspace || workspace_bytes < need) {
        set_error("warp_volume_coords_img: workspace %zu bytes < required %zu", workspace_bytes, need);
        return MPHIP_EWORKSPACE;
    }
    return warp_volume_gather(v, coords, out, out_range, (int *)workspace, (float *)img, true, B, C, D, H, W, (hipStream_t)stream);
}

// K3 with the coordinate pass already done (mphip_warp_coords): lets a caller look at the sample positions BEFORE the volume is
// produced (mphip_warp_sample_box -> mphip_conv3d_fwd_roi).  shared != 0: v is ONE volume [1,C,D,H,W] for all B coordinate sets.
extern "C" int mphip_warp_volume_dsum_coords(const float *v, const float *coords, float *out, int B, int C, int D, int H, int W, int shared,
                                             void *stream) {
    MPHIP_REQUIRE(v && coords && out, "warp_volume_dsum_coords: null pointer");
    MPHIP_REQUIRE(B > 0 && C > 0 && D > 0 && H > 0 && W > 0, "warp_volume_dsum_coords: bad dims");
    constexpr int CPB = 16;
    const int tiles = ((H + K3_TH - 1) / K3_TH) * ((W + K3_TW - 1) / K3_TW);
    hipLaunchKernelGGL(warp_gather_dsum_kernel<CPB>, dim3((unsigned)((size_t)B * tiles * cdiv(C, CPB))), dim3(256), 0, (hipStream_t)stream, v,
                       coords, out, B, C, D, H, W, shared ? (size_t)0 : (size_t)C * D * H * W);
    return check_launch("warp_volume_dsum_coords");
}

extern "C" int mphip_warp_volume_dsum(const float *v, const float *field, const float *lin_d, const float *lin_h,
                                      const float *lin_w, float *out, int B, int C, int D, int H, int W, int fD,
                                      int fH, int fW, void *workspace, size_t workspace_bytes, void *stream) {
    return warp_volume_dsum_impl("warp_volume_dsum", v, (size_t)C * D * H * W, field, lin_d, lin_h, lin_w, out, B, C, D, H, W, fD,
                                 fH, fW, workspace, workspace_bytes, stream);
}

extern "C" int mphip_warp_volume_dsum_shared(const float *v, const float *field, const float *lin_d, const float *lin_h,
                                             const float *lin_w, float *out, int B, int C, int D, int H, int W, int fD,
                                             int fH, int fW, void *workspace, size_t workspace_bytes, void *stream) {
    return warp_volume_dsum_impl("warp_volume_dsum_shared", v, 0, field, lin_d, lin_h, lin_w, out, B, C, D, H, W, fD, fH, fW,
                                 workspace, workspace_bytes, stream);
}

// =====================================================================================================
// K10 — backward of K1/K2/K3 (scope row f2).  Gradients as torch.autograd gives them for the reference's ops:
// F.grid_sample(bilinear, border, align_corners=True) wrt input and grid (ATen GridSampler backward: clipped
// coordinates pass no gradient), the align_corners=True resize of the field, torch.sum(dim=2), F.affine_grid and
// the align_corners=False resize of the flow field.
namespace mphip {

// Scatter pass (dv via hardware fp32 atomics — the reference's own CUDA backward is an atomicAdd scatter too — and the
// coordinate gradient of each channel slice): a workgroup owns a 4x16x16 tile of output voxels (4 per thread) and WB_CH
// channels.  With a smooth field the source voxels of the tile form a small box (as in K2): dv contributions are
// accumulated in an LDS image of that box (ds_add_f32) and flushed with ONE global atomic per box element, in
// coalesced rows — ~1.7 global atomics per output value instead of 8 scattered ones.  (Tried on top, both without gain:
// explicit ds_add_f32 instead of flat atomics, +-0 %; handing x1 contributions to the x-neighbour lane by DPP to halve the
// LDS atomics, -20 %; storing the box image to a per-tile scratch slot and summing covering boxes per dv element in a second
// pass instead of the atomic flush, -60 %, and still not bitwise reproducible because the ds_add_f32 order varies.)  A box that does not fit
// (wild field) falls back to direct global atomics for that tile.
constexpr int WB_CH = 8;
constexpr int WB_LDS = 16384;  // floats: 64 KB of accumulation image
constexpr int FBOX_INTS = 8;   // per-frame sample box (warp_frame_box_kernel): ox, oy, oz, ex, ey, ez, dense, -
template <bool DSUM>
__global__ void __launch_bounds__(256)
warp_bwd_tiled_kernel(const float *__restrict__ v, const float *__restrict__ coords, const float *__restrict__ dout,
                      float *__restrict__ dv_all, float *__restrict__ dcoords, const int *__restrict__ fbox, int B, int C, int D,
                      int H, int W) {
    __shared__ float img[WB_LDS];
    __shared__ int red[24];
    const int HW = H * W;
    const size_t vol = (size_t)D * HW;
    const int tiles_w = (W + 15) / 16, tiles_h = (H + 15) / 16, tiles_d = (D + 3) / 4;
    int bid = blockIdx.x;
    const int tw = bid % tiles_w; bid /= tiles_w;
    const int th = bid % tiles_h; bid /= tiles_h;
    const int td = bid % tiles_d;
    const int b = bid / tiles_d;
    const int ox = tw * 16 + (threadIdx.x & 15), oy = th * 16 + (threadIdx.x >> 4);
    const int c0 = blockIdx.y * WB_CH, cs = min(WB_CH, C - c0);
    // a frame whose samples all fall into one small box belongs to the warp_bwd_dense_* kernels (block-uniform)
    if (fbox && fbox[b * FBOX_INTS + 6]) return;
    float *const dv = dv_all;

    bool ok[4];
    int base[4], dxyz[4];
    float wx1[4], wy1[4], wz1[4], cxs[4], cys[4], czs[4];
    int x0s[4], y0s[4], z0s[4];
    int lx = INT_MAX, ly = INT_MAX, lz = INT_MAX, hx = 0, hy = 0, hz = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int oz = td * 4 + k;
        ok[k] = ox < W && oy < H && oz < D;
        const size_t t = (size_t)b * vol + (size_t)(ok[k] ? oz : 0) * HW + (ok[k] ? oy * W + ox : 0);
        const float cx = coords[t * 3], cy = coords[t * 3 + 1], cz = coords[t * 3 + 2];
        const int x0 = (int)floorf(cx), y0 = (int)floorf(cy), z0 = (int)floorf(cz);
        cxs[k] = cx; cys[k] = cy; czs[k] = cz;
        x0s[k] = x0; y0s[k] = y0; z0s[k] = z0;
        const bool vx = x0 + 1 < W, vy = y0 + 1 < H, vz = z0 + 1 < D;
        wx1[k] = vx ? cx - (float)x0 : 0.0f;  // the +1 corner outside: ATen skips it
        wy1[k] = vy ? cy - (float)y0 : 0.0f;
        wz1[k] = vz ? cz - (float)z0 : 0.0f;
        dxyz[k] = (vx ? 1 : 0) | (vy ? 2 : 0) | (vz ? 4 : 0);
        base[k] = (z0 * H + y0) * W + x0;
        if (ok[k]) {
            lx = min(lx, x0); ly = min(ly, y0); lz = min(lz, z0);
            hx = max(hx, x0); hy = max(hy, y0); hz = max(hz, z0);
        }
    }
    const Box bx = block_box(lx, ly, lz, hx, hy, hz, D, H, W, red);
    const int bvol = bx.ex * bx.ey * bx.ez;
    const bool staged = dv != nullptr && bvol * WB_CH <= WB_LDS && bvol > 0;  // block-uniform
    if (staged) {
        for (int i = threadIdx.x; i < bvol * cs; i += 256) img[i] = 0.0f;
        __syncthreads();
    }
    float gx[4] = {0.f, 0.f, 0.f, 0.f}, gy[4] = {0.f, 0.f, 0.f, 0.f}, gz[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < cs; ++c) {
        const size_t plane = (size_t)b * C + c0 + c;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!ok[k]) continue;
            const int oz = td * 4 + k;
            const float g = DSUM ? dout[plane * HW + oy * W + ox] : dout[plane * vol + (size_t)oz * HW + oy * W + ox];
            const bool vx = dxyz[k] & 1, vy = dxyz[k] & 2, vz = dxyz[k] & 4;
            const int dx = vx ? 1 : 0, dy = vy ? W : 0, dz = vz ? HW : 0;
            const float ax = (float)(x0s[k] + 1) - cxs[k], ay = (float)(y0s[k] + 1) - cys[k], az = (float)(z0s[k] + 1) - czs[k];
            const float bx1 = wx1[k], by1 = wy1[k], bz1 = wz1[k];
            if (dcoords) {
                const float *p = v + plane * vol + base[k];
                const float v000 = p[0], v100 = vx ? p[dx] : 0.0f, v010 = vy ? p[dy] : 0.0f, v110 = (vx && vy) ? p[dy + dx] : 0.0f;
                const float v001 = vz ? p[dz] : 0.0f, v101 = (vz && vx) ? p[dz + dx] : 0.0f;
                const float v011 = (vz && vy) ? p[dz + dy] : 0.0f, v111 = (vz && vy && vx) ? p[dz + dy + dx] : 0.0f;
                gx[k] += g * (((v100 - v000) * ay + (v110 - v010) * by1) * az + ((v101 - v001) * ay + (v111 - v011) * by1) * bz1);
                gy[k] += g * (((v010 - v000) * ax + (v110 - v100) * bx1) * az + ((v011 - v001) * ax + (v111 - v101) * bx1) * bz1);
                gz[k] += g * (((v001 - v000) * ax + (v101 - v100) * bx1) * ay + ((v011 - v010) * ax + (v111 - v110) * bx1) * by1);
            }
            if (dv) {
                float *q;
                int sx, sy, sz;
                if (staged) {
                    q = img + c * bvol + ((z0s[k] - bx.oz) * bx.ey + (y0s[k] - bx.oy)) * bx.ex + (x0s[k] - bx.ox);
                    sx = dx; sy = vy ? bx.ex : 0; sz = vz ? bx.ex * bx.ey : 0;
                } else {
                    q = dv + plane * vol + base[k];
                    sx = dx; sy = dy; sz = dz;
                }
                const float w00 = ay * az * g, w10 = by1 * az * g, w01 = ay * bz1 * g, w11 = by1 * bz1 * g;
                unsafeAtomicAdd(q, ax * w00);
                if (bx1 != 0.0f) unsafeAtomicAdd(q + sx, bx1 * w00);
                if (by1 != 0.0f) {
                    unsafeAtomicAdd(q + sy, ax * w10);
                    if (bx1 != 0.0f) unsafeAtomicAdd(q + sy + sx, bx1 * w10);
                }
                if (bz1 != 0.0f) {
                    unsafeAtomicAdd(q + sz, ax * w01);
                    if (bx1 != 0.0f) unsafeAtomicAdd(q + sz + sx, bx1 * w01);
                    if (by1 != 0.0f) {
                        unsafeAtomicAdd(q + sz + sy, ax * w11);
                        if (bx1 != 0.0f) unsafeAtomicAdd(q + sz + sy + sx, bx1 * w11);
                    }
                }
            }
        }
    }
    if (staged) {
        __syncthreads();
        const int exy = bx.ex * bx.ey;
        for (int i = threadIdx.x; i < bvol * cs; i += 256) {
            const float a = img[i];
            if (a == 0.0f) continue;
            const int c = i / bvol, e = i - c * bvol;
            const int z = e / exy, r2 = e - z * exy, y = r2 / bx.ex, x = r2 - y * bx.ex;
            unsafeAtomicAdd(dv + ((size_t)b * C + c0 + c) * vol + (size_t)(bx.oz + z) * HW + (bx.oy + y) * W + bx.ox + x, a);
        }
    }
    if (dcoords) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!ok[k]) continue;
            const size_t t = (size_t)b * vol + (size_t)(td * 4 + k) * HW + oy * W + ox;
            float *o = dcoords + ((size_t)blockIdx.y * B * vol + t) * 3;
            o[0] = (cxs[k] > 0.0f && cxs[k] < (float)(W - 1)) ? gx[k] : 0.0f;
            o[1] = (cys[k] > 0.0f && cys[k] < (float)(H - 1)) ? gy[k] : 0.0f;
            o[2] = (czs[k] > 0.0f && czs[k] < (float)(D - 1)) ? gz[k] : 0.0f;
        }
    }
}

// ---- the reference's own fields: every sample of a frame inside one small box ------------------------------------------
// apply_warping_field hands grid_sample coordinates of size ~[-2, 3] as if they were voxel indices, so
// after the border clip EVERY output voxel of a frame samples the low corner of the volume: floor indices in {0, 1(, 2, 3)}.
// dv is then non-zero in E^3 voxels per channel (E = 3..5) and each of them receives a contribution from all D*H*W outputs: in the
// tiled scatter above that is ~1000 same-address LDS atomics per box element and tile, fully serialised (1.18 ms per warp at
// B=4, 15 % of a training step).  For such frames dv is a plain reduction over the outputs,
//     dv[c][cell] = sum_o dout[c][o] * Wt[o][cell],      Wt[o][cell] = fz(cell.z - z0(o)) * fy(..) * fx(..)
// i.e. a [C x outputs] x [outputs x E^3] GEMM with exact fp32 products: it runs on v_mfma_f32_32x32x2_f32 (M = 32 channels,
// N = 32 box cells, K = 2 outputs).  A lane supplies dout of its channel (one 16-byte load per four k-steps) and computes the
// trilinear weight of ITS cell for the k-step's output (zero unless the cell is one of the output's 8 corners) — no atomics, no
// LDS traffic in the loop.  Every wave reduces its own range of outputs; a workgroup folds its 4 waves in LDS and writes one
// partial [C][cells]; warp_bwd_dense_fold_kernel sums the partials of a frame in a fixed order (deterministic, unlike the scatter)
// and stores the box into the zero-filled dv.  The per-frame box comes from a one-workgroup-per-frame pass over the coordinates
// (fbox[6] = the smallest E in 3..5 that holds the frame, 0 = none): frames that do not qualify keep the tiled scatter.
constexpr int DENSE_E_MIN = 3, DENSE_E_MAX = 5;
constexpr int DENSE_SEGS = 64;       // workgroups (partials) per frame and 96-channel block
constexpr int DENSE_MT = 3;          // 32-channel MFMA row tiles per workgroup
constexpr int DENSE_COLS = 128;      // column stride of a partial (>= 5^3)
__host__ __device__ constexpr int dense_ntiles(int E) { return E == 3 ? 1 : E == 4 ? 2 : 4; }  // 32-cell MFMA column tiles

__global__ void __launch_bounds__(1024)
warp_frame_box_kernel(const float *__restrict__ coords, int *__restrict__ fbox, int D, int H, int W, int allow_dense) {
    __shared__ int red[16 * 6];
    const int b = blockIdx.x;
    const size_t vol = (size_t)D * H * W;
    const float *cb = coords + (size_t)b * vol * 3;
    int lx = INT_MAX, ly = INT_MAX, lz = INT_MAX, hx = 0, hy = 0, hz = 0;
    auto take = [&](float x, float y, float z) {
        const int x0 = (int)floorf(x), y0 = (int)floorf(y), z0 = (int)floorf(z);
        lx = min(lx, x0); ly = min(ly, y0); lz = min(lz, z0);
        hx = max(hx, x0); hy = max(hy, y0); hz = max(hz, z0);
    };
    if (vol % 4 == 0 && ((uintptr_t)cb & 15) == 0) {
        // four voxels = three 16-byte loads of contiguous memory per thread and step, four steps in flight: the frame's 786 KB stream through
        // ONE workgroup at the CU's load rate (r04's strided dword loads, 24 per step: 55 us per call, the longest single item in front of
        // the demand-driven final_conv on a one-stream step)
        const float4 *c4 = reinterpret_cast<const float4 *>(cb);
        const size_t groups = vol / 4;
        for (size_t g0 = threadIdx.x; g0 < groups; g0 += 4 * 1024) {
            float4 q[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t g = min(g0 + (size_t)u * 1024, groups - 1);  // (a clamped duplicate changes no minimum / maximum)
#pragma unroll
                for (int k = 0; k < 3; ++k) q[u][k] = c4[g * 3 + k];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                take(q[u][0].x, q[u][0].y, q[u][0].z);
                take(q[u][0].w, q[u][1].x, q[u][1].y);
                take(q[u][1].z, q[u][1].w, q[u][2].x);
                take(q[u][2].y, q[u][2].z, q[u][2].w);
            }
        }
    } else {
        for (size_t t0 = threadIdx.x; t0 < vol; t0 += 8 * 1024) {
            float c[8][3];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const size_t t = min(t0 + (size_t)u * 1024, vol - 1);  // (a clamped duplicate changes no minimum / maximum)
                c[u][0] = cb[t * 3]; c[u][1] = cb[t * 3 + 1]; c[u][2] = cb[t * 3 + 2];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) take(c[u][0], c[u][1], c[u][2]);
        }
    }
    lx = wave_min(lx); ly = wave_min(ly); lz = wave_min(lz);
    hx = wave_max(hx); hy = wave_max(hy); hz = wave_max(hz);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[wave * 6 + 0] = lx; red[wave * 6 + 1] = ly; red[wave * 6 + 2] = lz;
        red[wave * 6 + 3] = hx; red[wave * 6 + 4] = hy; red[wave * 6 + 5] = hz;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) {
            lx = min(lx, red[w * 6 + 0]); ly = min(ly, red[w * 6 + 1]); lz = min(lz, red[w * 6 + 2]);
            hx = max(hx, red[w * 6 + 3]); hy = max(hy, red[w * 6 + 4]); hz = max(hz, red[w * 6 + 5]);
        }
        const int ex = min(hx + 1, W - 1) - lx + 1, ey = min(hy + 1, H - 1) - ly + 1, ez = min(hz + 1, D - 1) - lz + 1;
        const int e = max(max(ex, ey), max(ez, DENSE_E_MIN));
        // (the GEMM walks the outputs in aligned groups of 4: 16-byte loads of dout)
        const bool shape_ok = (H * W) % 4 == 0 && vol >= 32;
        int *o = fbox + b * FBOX_INTS;
        o[0] = lx; o[1] = ly; o[2] = lz; o[3] = ex; o[4] = ey; o[5] = ez;
        o[6] = (e <= DENSE_E_MAX && shape_ok && allow_dense) ? e : 0;
        o[7] = 0;
    }
}

template <bool DSUM, int E>
__global__ void __launch_bounds__(256)
warp_bwd_dense_dv_kernel(const float *__restrict__ coords, const float *__restrict__ dout, float *__restrict__ partial,
                         const int *__restrict__ fbox, int C, int D, int H, int W) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    constexpr int NT = dense_ntiles(E), MT = DENSE_MT;
    const int b = blockIdx.z;
    const int *fb = fbox + b * FBOX_INTS;
    if (fb[6] != E) return;  // block-uniform: another instantiation (or the tiled scatter) owns this frame
    const int ox = fb[0], oy = fb[1], oz = fb[2];
    const int HW = H * W;
    const size_t vol = (size_t)D * HW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int cblk = blockIdx.y * (MT * 32);
    // this wave's outputs: [t0, t1), walked 32 at a time: lane half h takes outputs +16h .. +16h+15, one per k-step (the order of
    // the k index is free).  One iteration = 16 k-steps x MT x NT MFMAs (1.5-6 us of matrix time), with the next iteration's
    // loads (4 x 16 B of dout per lane and row tile, one coordinate triple per lane) in flight underneath.
    const size_t per_wave = (vol / 32 + DENSE_SEGS * 4 - 1) / (DENSE_SEGS * 4) * 32;
    const size_t t0 = min(vol, ((size_t)blockIdx.x * 4 + wave) * per_wave), t1 = min(vol, t0 + per_wave);
    // LDS table of the current 32 outputs: per output the dense per-axis weight vectors fx[0..E), fy[0..E), fz[0..E) (a slot of
    // zeros at [15]); the weight of (output, cell) is one product of three table reads, no select chains in the k loop
    __shared__ __attribute__((aligned(16))) float tab_all[4][32][16];
    float (*tab)[16] = tab_all[wave];
    int offx[NT], offy[NT], offz[NT];  // table slots of this lane's column (box cell x, y, z) in every column tile
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int cell = n * 32 + col;
        offx[n] = cell % E; offy[n] = 5 + (cell / E) % E;
        offz[n] = cell < E * E * E ? 10 + cell / (E * E) : 15;  // padding columns read the zero slot
    }
    const float *gp[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) gp[m] = dout + ((size_t)b * C + min(cblk + m * 32 + col, C - 1)) * (DSUM ? (size_t)HW : vol);
    const float *cb = coords + (size_t)b * vol * 3;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    float4 gv[MT][4], gv_n[MT][4];
    float cx_l, cy_l, cz_l, cx_n, cy_n, cz_n;  // the coordinates of output t + lane % 32 (lanes 32..63 mirror 0..31)
    // branch-free and without selects on the loaded values (a conditional load is waited for at the end of its block, a select
    // right after the load — in both cases before the MFMAs the load should hide under): addresses are clamped into the frame;
    // outputs past the end of the range get an all-zero table row below, i.e. weight 0 (rows of channels >= C are never read back)
    auto load = [&](size_t t, float4 (*g)[4], float &lx_, float &ly_, float &lz_) {
        const size_t o = t + 16 * half;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const size_t oq = min(o + 4 * q, vol - 4);
            const size_t gi = DSUM ? oq % (size_t)HW : oq;  // (H*W % 4 == 0: a group of 4 never leaves its plane)
#pragma unroll
            for (int m = 0; m < MT; ++m) g[m][q] = *reinterpret_cast<const float4 *>(gp[m] + gi);
        }
        const size_t oc = min(t + (size_t)col, vol - 1);
        lx_ = cb[oc * 3]; ly_ = cb[oc * 3 + 1]; lz_ = cb[oc * 3 + 2];
    };
    if (t0 < t1) load(t0, gv, cx_l, cy_l, cz_l);
    for (size_t t = t0; t < t1; t += 32) {
        load(min(t + 32, vol - 32), gv_n, cx_n, cy_n, cz_n);  // (the last iteration's prefetch is a discarded re-read)
        {   // this lane's output (t + col): the same weights as the tiled kernel — (x0+1) - cx on the floor corner, cx - x0 on
            // the +1 corner unless it is outside the volume
            const int x0 = (int)floorf(cx_l), y0 = (int)floorf(cy_l), z0 = (int)floorf(cz_l);
            const float ax = (float)(x0 + 1) - cx_l, ay = (float)(y0 + 1) - cy_l, az = (float)(z0 + 1) - cz_l;
            const float bx1 = x0 + 1 < W ? cx_l - (float)x0 : 0.0f, by1 = y0 + 1 < H ? cy_l - (float)y0 : 0.0f,
                        bz1 = z0 + 1 < D ? cz_l - (float)z0 : 0.0f;
            const int ix = x0 - ox, iy = y0 - oy;
            const int iz = t + col < t1 ? z0 - oz : -2;  // past the end of this wave's range: no z slot matches, weight 0
            float f[16];
#pragma unroll
            for (int e = 0; e < 5; ++e) {
                f[e] = e == ix ? ax : e == ix + 1 ? bx1 : 0.0f;
                f[5 + e] = e == iy ? ay : e == iy + 1 ? by1 : 0.0f;
                f[10 + e] = e == iz ? az : e == iz + 1 ? bz1 : 0.0f;
            }
            f[15] = 0.0f;
            __builtin_amdgcn_wave_barrier();  // (every lane is past its reads of the previous table: one wave, program order)
            if (half == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<float4 *>(&tab[col][4 * q]) = make_float4(f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float *row = tab[16 * half + i];
            float w[NT];
#pragma unroll
#ifndef MPHIP_DENSE_ABL_NOTAB
            for (int n = 0; n < NT; ++n) w[n] = row[offz[n]] * row[offy[n]] * row[offx[n]];
#else
            for (int n = 0; n < NT; ++n) w[n] = cx_l + (float)(n + i);
#endif
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float4 q4 = gv[m][i >> 2];
                const float gmi = (i & 3) == 0 ? q4.x : (i & 3) == 1 ? q4.y : (i & 3) == 2 ? q4.z : q4.w;
#pragma unroll
#ifndef MPHIP_DENSE_ABL_NOMFMA
                for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(gmi, w[n], acc[m][n], 0, 0, 0);
#else
                for (int n = 0; n < NT; ++n) acc[m][n][(i + n) & 15] += gmi * w[n];
#endif
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) gv[m][q] = gv_n[m][q];
        cx_l = cx_n; cy_l = cy_n; cz_l = cz_n;
    }
    // fold the 4 waves pairwise through LDS with plain stores / loads, [value][lane] (conflict-free) — ds_add_f32 costs ~4 cycles
    // per LANE: 192 of them per lane were 75 % of this kernel's time — then wave 0 writes the workgroup's partial [MT*32][DENSE_COLS]
    __shared__ float xch[2][MT * NT * 16][64];
#pragma unroll
    for (int step = 0; step < 2; ++step) {
        const int senders_from = step == 0 ? 2 : 1, nsend = step == 0 ? 2 : 1;  // waves 2,3 -> 0,1 ; then wave 1 -> 0
        if (wave >= senders_from && wave < senders_from + nsend) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) xch[wave - senders_from][(m * NT + n) * 16 + r][lane] = acc[m][n][r];
        }
        __syncthreads();
        if (wave < nsend) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][n][r] += xch[wave][(m * NT + n) * 16 + r][lane];
        }
        __syncthreads();
    }
    if (wave == 0) {
        float *pw = partial + (((size_t)b * gridDim.y + blockIdx.y) * DENSE_SEGS + blockIdx.x) * (MT * 32 * DENSE_COLS);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;  // D layout: row (channel), column = lane & 31 (cell)
                    pw[row * DENSE_COLS + n * 32 + col] = acc[m][n][r];
                }
    }
}

// dv[b][c][box cell] = sum over the frame's DENSE_SEGS partials, in index order (dv is zero-filled: only the box is written)
__global__ void __launch_bounds__(256)
warp_bwd_dense_fold_kernel(const float *__restrict__ partial, float *__restrict__ dv, const int *__restrict__ fbox, int C, int D,
                           int H, int W, int cblocks) {
    const int b = blockIdx.y;
    const int *fb = fbox + b * FBOX_INTS;
    const int E = fb[6];
    if (!E) return;
    const int cells = E * E * E;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= C * cells) return;
    const int c = t / cells, cell = t % cells;
    const int x = cell % E, y = (cell / E) % E, z = cell / (E * E);
    if (x >= fb[3] || y >= fb[4] || z >= fb[5]) return;  // outside the (border-clamped) box: its weights are all zero
    const float *p = partial + ((size_t)b * cblocks + c / (DENSE_MT * 32)) * DENSE_SEGS * (DENSE_MT * 32 * DENSE_COLS) +
                     (size_t)(c % (DENSE_MT * 32)) * DENSE_COLS + cell;
    float a = 0.0f;
    for (int sgm = 0; sgm < DENSE_SEGS; ++sgm) a += p[(size_t)sgm * (DENSE_MT * 32 * DENSE_COLS)];
    const int HW = H * W;
    dv[((size_t)b * C + c) * D * HW + (size_t)(fb[2] + z) * HW + (fb[1] + y) * W + fb[0] + x] = a;
}

// Coordinate gradient of the same frames.  d out / d coord is linear in the 8 corner values, so per output voxel
//     S_k = sum_c dout[c][o] * v[c][corner_k(o)]       (8 FMAs per channel, the corners read from an LDS copy of the box)
// and the trilinear derivative formulas are applied ONCE to the eight S_k instead of once per channel.  One slab of gradients
// (the tiled kernel writes one per 8-channel slice for the resize adjoint to sum), no re-read of v from HBM.
constexpr int DENSE_DC_CH = 96;   // channels per LDS box image (48 KB at E = 5)
template <bool DSUM>
__global__ void __launch_bounds__(256)
warp_bwd_dense_dcoords_kernel(const float *__restrict__ v, const float *__restrict__ coords, const float *__restrict__ dout,
                              float *__restrict__ dcoords, const int *__restrict__ fbox, int C, int D, int H, int W) {
    __shared__ float vbox[DENSE_DC_CH * DENSE_E_MAX * DENSE_E_MAX * DENSE_E_MAX];
    const int b = blockIdx.y;
    const int *fb = fbox + b * FBOX_INTS;
    const int E = fb[6];
    if (!E) return;  // block-uniform: the tiled kernel owns this frame
    const int ox = fb[0], oy = fb[1], oz = fb[2], ex = fb[3], ey = fb[4], ez = fb[5];
    const int cells = E * E * E;
    const int HW = H * W;
    const size_t vol = (size_t)D * HW;
    const size_t seg_len = ((vol + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    const size_t t_begin = blockIdx.x * seg_len;
    constexpr int OPT = 4;  // outputs per thread
    float S[OPT][8];
#pragma unroll
    for (int i = 0; i < OPT; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k) S[i][k] = 0.0f;
    int cbase[OPT], dx[OPT], dy[OPT], dz[OPT];
    size_t gi[OPT];
    bool ok[OPT];
#pragma unroll
    for (int i = 0; i < OPT; ++i) {
        const size_t t = t_begin + (size_t)i * 256 + threadIdx.x;
        ok[i] = t < min(vol, t_begin + seg_len);
        const size_t tc = ok[i] ? t : 0;
        const float *cp = coords + ((size_t)b * vol + tc) * 3;
        const int x0 = (int)floorf(cp[0]), y0 = (int)floorf(cp[1]), z0 = (int)floorf(cp[2]);
        dx[i] = x0 + 1 < W ? 1 : 0;
        dy[i] = y0 + 1 < H ? E : 0;
        dz[i] = z0 + 1 < D ? E * E : 0;
        cbase[i] = ((z0 - oz) * E + (y0 - oy)) * E + (x0 - ox);
        gi[i] = DSUM ? tc % (size_t)HW : tc;
    }
    for (int c0 = 0; c0 < C; c0 += DENSE_DC_CH) {
        const int cs = min(DENSE_DC_CH, C - c0);
        __syncthreads();
        for (int i0 = threadIdx.x; i0 < cs * cells; i0 += 8 * 256) {  // eight loads in flight per thread (47 dependent ones otherwise)
            float val[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(i0 + u * 256, cs * cells - 1);
                const int c = i / cells, cell = i - c * cells;
                const int x = cell % E, y = (cell / E) % E, z = cell / (E * E);
                const bool in = x < ex && y < ey && z < ez;
                const float got = v[((size_t)b * C + c0 + c) * vol + (size_t)(oz + min(z, ez - 1)) * HW + (oy + min(y, ey - 1)) * W + ox + min(x, ex - 1)];
                val[u] = in ? got : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + u * 256 < cs * cells) vbox[i0 + u * 256] = val[u];
        }
        __syncthreads();
        const float *gp = dout + ((size_t)b * C + c0) * (DSUM ? (size_t)HW : vol);
#pragma unroll 4
        for (int c = 0; c < cs; ++c) {
            const float *vb = vbox + c * cells;
#pragma unroll
            for (int i = 0; i < OPT; ++i) {
                const float g = ok[i] ? gp[(size_t)c * (DSUM ? (size_t)HW : vol) + gi[i]] : 0.0f;
                const float *q = vb + cbase[i];
                S[i][0] = fmaf(g, q[0], S[i][0]);
                S[i][1] = fmaf(g, q[dx[i]], S[i][1]);
                S[i][2] = fmaf(g, q[dy[i]], S[i][2]);
                S[i][3] = fmaf(g, q[dy[i] + dx[i]], S[i][3]);
                S[i][4] = fmaf(g, q[dz[i]], S[i][4]);
                S[i][5] = fmaf(g, q[dz[i] + dx[i]], S[i][5]);
                S[i][6] = fmaf(g, q[dz[i] + dy[i]], S[i][6]);
                S[i][7] = fmaf(g, q[dz[i] + dy[i] + dx[i]], S[i][7]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < OPT; ++i) {
        if (!ok[i]) continue;
        const size_t t = t_begin + (size_t)i * 256 + threadIdx.x;
        const float *cp = coords + ((size_t)b * vol + t) * 3;
        const float cx = cp[0], cy = cp[1], cz = cp[2];
        const int x0 = (int)floorf(cx), y0 = (int)floorf(cy), z0 = (int)floorf(cz);
        const bool vx = dx[i] != 0, vy = dy[i] != 0, vz = dz[i] != 0;
        // corners outside the volume: ATen skips them (value 0, weight 0) — the same rule as the tiled kernel
        const float ax = (float)(x0 + 1) - cx, ay = (float)(y0 + 1) - cy, az = (float)(z0 + 1) - cz;
        const float bx1 = vx ? cx - (float)x0 : 0.0f, by1 = vy ? cy - (float)y0 : 0.0f, bz1 = vz ? cz - (float)z0 : 0.0f;
        const float v000 = S[i][0], v100 = vx ? S[i][1] : 0.0f, v010 = vy ? S[i][2] : 0.0f, v110 = (vx && vy) ? S[i][3] : 0.0f;
        const float v001 = vz ? S[i][4] : 0.0f, v101 = (vz && vx) ? S[i][5] : 0.0f, v011 = (vz && vy) ? S[i][6] : 0.0f,
                    v111 = (vz && vy && vx) ? S[i][7] : 0.0f;
        const float gx = ((v100 - v000) * ay + (v110 - v010) * by1) * az + ((v101 - v001) * ay + (v111 - v011) * by1) * bz1;
        const float gy = ((v010 - v000) * ax + (v110 - v100) * bx1) * az + ((v011 - v001) * ax + (v111 - v101) * bx1) * bz1;
        const float gz = ((v001 - v000) * ax + (v101 - v100) * bx1) * ay + ((v011 - v010) * ax + (v111 - v110) * bx1) * by1;
        float *o = dcoords + ((size_t)b * vol + t) * 3;  // slab 0
        o[0] = (cx > 0.0f && cx < (float)(W - 1)) ? gx : 0.0f;
        o[1] = (cy > 0.0f && cy < (float)(H - 1)) ? gy : 0.0f;
        o[2] = (cz > 0.0f && cz < (float)(D - 1)) ? gz : 0.0f;
    }
}

template <bool DSUM, int E>
static void launch_dense_dv(const float *coords, const float *dout, float *partial, const int *fbox, int B, int C, int D, int H,
                            int W, hipStream_t s) {
    hipLaunchKernelGGL((warp_bwd_dense_dv_kernel<DSUM, E>), dim3(DENSE_SEGS, cdiv(C, DENSE_MT * 32), B), dim3(256), 0, s, coords,
                       dout, partial, fbox, C, D, H, W);
}

template <bool ALIGN>
__device__ __forceinline__ void adj_bounds(int i, int in, int out, int &lo, int &hi) {
    // outputs whose source interval [i0, i1] can contain input index i (padded by one for rounding; weights decide)
    float a, b2;
    if (ALIGN) {
        const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.0f;
        if (!(scale > 0.0f)) { lo = 0; hi = out - 1; return; }
        a = ((float)i - 1.0f) / scale;
        b2 = ((float)i + 1.0f) / scale;
    } else {
        const float scale = (float)in / (float)out;
        a = ((float)i - 0.5f) / scale - 0.5f;
        b2 = ((float)i + 1.5f) / scale - 0.5f;
    }
    lo = max(0, (int)floorf(a) - 1);
    hi = min(out - 1, (int)ceilf(b2) + 1);
    if (!ALIGN && i == 0) lo = 0;  // src is clamped at 0: every o below the first source lands on i0 = 0
}
template <bool ALIGN>
__device__ __forceinline__ float adj_w(int o, int i, int in, int out) {
    const SrcIdx s = src_index<ALIGN>(o, in, out);
    float w = 0.0f;
    if (s.i0 == i) w += s.l0;
    if (s.i1 == i) w += s.l1;
    return w;
}

// adjoint of a trilinear resize [B,C,iD,iH,iW] -> [B,C,oD,oH,oW]: gin[i] = sum_o w(o -> i) * sum_slabs gout[o]
// (gather form, deterministic).  interleaved: gout is [slab][B][oVol][C] (the coordinate-gradient layout of
// warp_bwd_kernel) instead of [slab][B][C][oVol].
template <bool ALIGN>
__global__ void __launch_bounds__(256)
resize_trilinear_adjoint_kernel(const float *__restrict__ gout, float *__restrict__ gin, int B, int C, int iD, int iH, int iW,
                                int oD, int oH, int oW, int slabs_all, int interleaved, const int *__restrict__ fbox) {
    const size_t ivol = (size_t)iD * iH * iW, ovol = (size_t)oD * oH * oW;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)B * C * ivol) return;
    const int iw = (int)(t % iW);
    size_t r = t / iW;
    const int ih = (int)(r % iH);
    r /= iH;
    const int id = (int)(r % iD);
    r /= iD;
    const int ch = (int)(r % C), b = (int)(r / C);
    const int slabs = (fbox && fbox[b * FBOX_INTS + 6]) ? 1 : slabs_all;  // a dense frame's coordinate gradient is one slab
    int dlo, dhi, hlo, hhi, wlo, whi;
    adj_bounds<ALIGN>(id, iD, oD, dlo, dhi);
    adj_bounds<ALIGN>(ih, iH, oH, hlo, hhi);
    adj_bounds<ALIGN>(iw, iW, oW, wlo, whi);
    const size_t slab_stride = (size_t)B * C * ovol;
    float acc = 0.0f;
    for (int od = dlo; od <= dhi; ++od) {
        const float wd = adj_w<ALIGN>(od, id, iD, oD);
        if (wd == 0.0f) continue;
        float pl = 0.0f;
        for (int oh = hlo; oh <= hhi; ++oh) {
            const float wh = adj_w<ALIGN>(oh, ih, iH, oH);
            if (wh == 0.0f) continue;
            float rs = 0.0f;
            for (int ow = wlo; ow <= whi; ++ow) {
                const float ww = adj_w<ALIGN>(ow, iw, iW, oW);
                if (ww == 0.0f) continue;
                const size_t o = ((size_t)od * oH + oh) * oW + ow;
                const size_t idx = interleaved ? ((size_t)b * ovol + o) * C + ch : ((size_t)b * C + ch) * ovol + o;
                float g = gout[idx];
                for (int s = 1; s < slabs; ++s) g += gout[(size_t)s * slab_stride + idx];
                rs += ww * g;
            }
            pl += wh * rs;
        }
        acc += wd * pl;
    }
    gin[t] = acc;
}

// one axis of the same adjoint (the trilinear resize is separable): gin[outer][i][inner] = sum_o w(o -> i) * gout[outer][o][inner].
// Three of these replace the 3-D gather when the candidate box is large (16 -> 64 upsampling: ~8^3 outputs per input).
template <bool ALIGN>
__global__ void __launch_bounds__(256)
resize_adjoint_axis_kernel(const float *__restrict__ gout, float *__restrict__ gin, size_t outer, int in_len, int out_len,
                           size_t inner) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= outer * in_len * inner) return;
    const size_t q = t % inner;
    const int i = (int)((t / inner) % in_len);
    const size_t o_ = t / (inner * in_len);
    int lo, hi;
    adj_bounds<ALIGN>(i, in_len, out_len, lo, hi);
    const float *p = gout + (o_ * out_len) * inner + q;
    float acc = 0.0f;
    for (int o = lo; o <= hi; ++o) {
        const float w = adj_w<ALIGN>(o, i, in_len, out_len);
        if (w != 0.0f) acc += w * p[(size_t)o * inner];
    }
    gin[t] = acc;
}

// dtheta[b][j][k] = sum_p dw[b][j][p] * (x_p, y_p, z_p, 1)[k]   (F.affine_grid backward); partial sums per chunk
constexpr int TG_CHUNK = 8192;
__global__ void __launch_bounds__(256)
theta_grad_partial_kernel(const float *__restrict__ dw, const float *__restrict__ base, double *__restrict__ partial, int G,
                          int chunks) {
    const int bj = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    const size_t vol = (size_t)G * G * G;
    const float *p = dw + (size_t)bj * vol;
    const size_t begin = (size_t)chunk * TG_CHUNK, end = min(vol, begin + TG_CHUNK);
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (size_t i = begin + threadIdx.x; i < end; i += 256) {
        const float g = p[i];
        const int w = (int)(i % G), h = (int)((i / G) % G), d = (int)(i / ((size_t)G * G));
        s[0] += (double)(g * base[w]);
        s[1] += (double)(g * base[h]);
        s[2] += (double)(g * base[d]);
        s[3] += (double)g;
    }
    __shared__ double red[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) s[k] += __shfl_xor(s[k], sft, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = s[k];
    }
    __syncthreads();
    if (threadIdx.x < 4) partial[(size_t)blockIdx.x * 4 + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ void theta_grad_finalize_kernel(const double *__restrict__ partial, float *__restrict__ dtheta, int n, int chunks) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // over B*3*4
    if (i >= n) return;
    const int bj = i / 4, k = i % 4;
    double a = 0.0;
    for (int c = 0; c < chunks; ++c) a += partial[((size_t)bj * chunks + c) * 4 + k];
    dtheta[i] = (float)a;
}

// backward of rt_theta_kernel: theta = rows 0..2 of A (or of inv(A)), A = [Rx*Ry*Rz | t; 0 0 0 1], angles in degrees.
// d(inv A) -> dA = -M^T dM M^T with M = inv(A); then the product rule through the three axis rotations.
__global__ void rt_theta_bwd_kernel(const float *__restrict__ rot, const float *__restrict__ tr, const float *__restrict__ dtheta,
                                    float *__restrict__ drot, float *__restrict__ dtr, int B, int invert) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double k = 0.017453292519943295;
    const double ra = (double)(rot[b * 3] * 0.017453292519943295f), rb = (double)(rot[b * 3 + 1] * 0.017453292519943295f),
                 rg = (double)(rot[b * 3 + 2] * 0.017453292519943295f);
    const double ca = cos(ra), sa = sin(ra), cb = cos(rb), sb = sin(rb), cg = cos(rg), sg = sin(rg);
    const double Rx[3][3] = {{1, 0, 0}, {0, ca, -sa}, {0, sa, ca}};
    const double Ry[3][3] = {{cb, 0, sb}, {0, 1, 0}, {-sb, 0, cb}};
    const double Rz[3][3] = {{cg, -sg, 0}, {sg, cg, 0}, {0, 0, 1}};
    double YZ[3][3], XY[3][3], R[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0, u = 0.0;
            for (int q = 0; q < 3; ++q) { s += Ry[i][q] * Rz[q][j]; u += Rx[i][q] * Ry[q][j]; }
            YZ[i][j] = s;
            XY[i][j] = u;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int q = 0; q < 3; ++q) s += Rx[i][q] * YZ[q][j];
            R[i][j] = s;
        }
    double dA[3][4];  // gradient wrt the top three rows of A
    if (invert) {
        // A rigid: inv(A) = [R^T | -R^T t]; written through the general identity dA = -M^T dM M^T (dM's last row is 0)
        double M[4][4];
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) M[i][j] = R[j][i];
            double s = 0.0;
            for (int q = 0; q < 3; ++q) s += R[q][i] * (double)tr[b * 3 + q];
            M[i][3] = -s;
        }
        M[3][0] = M[3][1] = M[3][2] = 0.0;
        M[3][3] = 1.0;
        double T[4][4];  // T = M^T dM  (dM rows 0..2 = dtheta, row 3 = 0)
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double s = 0.0;
                for (int q = 0; q < 3; ++q) s += M[q][i] * (double)dtheta[(b * 3 + q) * 4 + j];
                T[i][j] = s;
            }
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) {
                double s = 0.0;
                for (int q = 0; q < 4; ++q) s += T[i][q] * M[j][q];
                dA[i][j] = -s;
            }
    } else {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) dA[i][j] = (double)dtheta[(b * 3 + i) * 4 + j];
    }
    for (int i = 0; i < 3; ++i) dtr[b * 3 + i] = (float)dA[i][3];
    // R = Rx * (Ry * Rz):  dRx = dR (YZ)^T,  dRy = Rx^T dR Rz^T,  dRz = (XY)^T dR
    double dRx[3][3], dRy[3][3], dRz[3][3], tmp[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0, u = 0.0, w = 0.0;
            for (int q = 0; q < 3; ++q) {
                s += dA[i][q] * YZ[j][q];
                u += Rx[q][i] * dA[q][j];
                w += XY[q][i] * dA[q][j];
            }
            dRx[i][j] = s;
            tmp[i][j] = u;
            dRz[i][j] = w;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int q = 0; q < 3; ++q) s += tmp[i][q] * Rz[j][q];
            dRy[i][j] = s;
        }
    const double da = dRx[1][1] * -sa + dRx[1][2] * -ca + dRx[2][1] * ca + dRx[2][2] * -sa;
    const double db = dRy[0][0] * -sb + dRy[0][2] * cb + dRy[2][0] * -cb + dRy[2][2] * -sb;
    const double dg = dRz[0][0] * -sg + dRz[0][1] * -cg + dRz[1][0] * cg + dRz[1][1] * -sg;
    drot[b * 3] = (float)(da * k);
    drot[b * 3 + 1] = (float)(db * k);
    drot[b * 3 + 2] = (float)(dg * k);
}

constexpr int WARP_BWD_CPB = WB_CH;  // channels per slice of the scatter pass

}  // namespace mphip

extern "C" int mphip_warp_coords(const float *field, const float *lin_d, const float *lin_h, const float *lin_w, float *coords,
                                 int B, int D, int H, int W, int fD, int fH, int fW, void *stream) {
    MPHIP_REQUIRE(field && lin_d && lin_h && lin_w && coords, "warp_coords: null pointer");
    MPHIP_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && fD > 0 && fH > 0 && fW > 0, "warp_coords: bad dims");
    return launch_coords(field, lin_d, lin_h, lin_w, coords, nullptr, B, D, H, W, fD, fH, fW, (hipStream_t)stream);
}

// Per frame, the box of source voxels the samples of `coords` [B,D,H,W,3] touch (all 8 trilinear corners, zero-weight ones
// included): box[b*8 ..] = {lx, ly, lz, ex, ey, ez, -, -} — origin and extent in voxels.  One workgroup per frame.
extern "C" int mphip_warp_sample_box(const float *coords, int *box, int B, int D, int H, int W, void *stream) {
    MPHIP_REQUIRE(coords && box, "warp_sample_box: null pointer");
    MPHIP_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, "warp_sample_box: bad dims");
    hipLaunchKernelGGL(warp_frame_box_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, coords, box, D, H, W, 0);
    return check_launch("warp_sample_box");
}

extern "C" size_t mphip_warp_volume_bwd_workspace_bytes(int B, int C, int D, int H, int W) {
    if (B <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    const size_t groups = (size_t)cdiv(C, WARP_BWD_CPB);
    const size_t dense = (size_t)B * cdiv(C, DENSE_MT * 32) * DENSE_SEGS * (DENSE_MT * 32 * DENSE_COLS) * sizeof(float) +
                         (size_t)B * FBOX_INTS * sizeof(int);  // partials of the dense dv path + per-frame sample boxes
    return (size_t)B * D * H * W * 3 * sizeof(float) * (1 + groups) + dense;
}

extern "C" int mphip_warp_volume_bwd(const float *v, const float *field, const float *lin_d, const float *lin_h,
                                     const float *lin_w, const float *dout, float *dv, float *dfield, int B, int C, int D,
                                     int H, int W, int fD, int fH, int fW, int dsum, void *workspace, size_t workspace_bytes,
                                     void *stream) {
    int rc = check_warp_args("warp_volume_bwd", v, field, lin_d, lin_h, lin_w, dout, B, C, D, H, W, fD, fH, fW);
    if (rc) return rc;
    MPHIP_REQUIRE(dv || dfield, "warp_volume_bwd: nothing to compute (dv and dfield are both NULL)");
    const size_t need = mphip_warp_volume_bwd_workspace_bytes(B, C, D, H, W);
    if (!workspace || workspace_bytes < need) {
        set_error("warp_volume_bwd: workspace %zu bytes < required %zu", workspace_bytes, need);
        return MPHIP_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t nvox = (size_t)B * D * H * W;
    float *coords = (float *)workspace, *dcoords = coords + nvox * 3;
    rc = launch_coords(field, lin_d, lin_h, lin_w, coords, nullptr, B, D, H, W, fD, fH, fW, s);
    if (rc) return rc;
    if (dv) {
        const size_t bytes = (size_t)B * C * D * H * W * sizeof(float);
        MPHIP_REQUIRE(bytes % 16 == 0 && ((uintptr_t)dv & 15) == 0, "warp_volume_bwd: dv must be 16-byte aligned / sized");
        zero_fill(dv, bytes, s);  // the scatter pass accumulates with atomics
    }
    const int groups = cdiv(C, WARP_BWD_CPB);
    // frames whose samples all sit in one small box (the reference's own fields) take the dense kernels: dv as a GEMM over the
    // outputs, the coordinate gradient from an LDS image of the box; every other frame the tiled scatter below
    float *partial = dcoords + nvox * 3 * groups;
    const int cblocks = cdiv(C, DENSE_MT * 32);
    int *fbox = (int *)(partial + (size_t)B * cblocks * DENSE_SEGS * (DENSE_MT * 32 * DENSE_COLS));
    const char *no_dense = getenv("MPHIP_WARP_BWD_DENSE");  // "0": every frame through the tiled scatter (tests: dense == tiled)
    hipLaunchKernelGGL(warp_frame_box_kernel, dim3(B), dim3(1024), 0, s, (const float *)coords, fbox, D, H, W,
                       (no_dense && no_dense[0] == '0') ? 0 : 1);
    if (dv) {
        if (dsum) {
            launch_dense_dv<true, 3>(coords, dout, partial, fbox, B, C, D, H, W, s);
            launch_dense_dv<true, 4>(coords, dout, partial, fbox, B, C, D, H, W, s);
            launch_dense_dv<true, 5>(coords, dout, partial, fbox, B, C, D, H, W, s);
        } else {
            launch_dense_dv<false, 3>(coords, dout, partial, fbox, B, C, D, H, W, s);
            launch_dense_dv<false, 4>(coords, dout, partial, fbox, B, C, D, H, W, s);
            launch_dense_dv<false, 5>(coords, dout, partial, fbox, B, C, D, H, W, s);
        }
        hipLaunchKernelGGL(warp_bwd_dense_fold_kernel, dim3(cdiv(C * 125, 256), B), dim3(256), 0, s, (const float *)partial, dv,
                           (const int *)fbox, C, D, H, W, cblocks);
    }
    if (dfield) {
        const dim3 dgrid((unsigned)cdiv((size_t)D * H * W, 1024), B);  // 4 outputs per thread
        if (dsum)
            hipLaunchKernelGGL(warp_bwd_dense_dcoords_kernel<true>, dgrid, dim3(256), 0, s, v, (const float *)coords, dout, dcoords,
                               (const int *)fbox, C, D, H, W);
        else
            hipLaunchKernelGGL(warp_bwd_dense_dcoords_kernel<false>, dgrid, dim3(256), 0, s, v, (const float *)coords, dout, dcoords,
                               (const int *)fbox, C, D, H, W);
    }
    dim3 grid((unsigned)((size_t)B * cdiv(D, 4) * cdiv(H, 16) * cdiv(W, 16)), groups);
    if (dsum)
        hipLaunchKernelGGL(warp_bwd_tiled_kernel<true>, grid, dim3(256), 0, s, v, (const float *)coords, dout, dv,
                           dfield ? dcoords : nullptr, (const int *)fbox, B, C, D, H, W);
    else
        hipLaunchKernelGGL(warp_bwd_tiled_kernel<false>, grid, dim3(256), 0, s, v, (const float *)coords, dout, dv,
                           dfield ? dcoords : nullptr, (const int *)fbox, B, C, D, H, W);
    if (dfield) {
        const size_t nf = (size_t)B * 3 * fD * fH * fW;
        hipLaunchKernelGGL(resize_trilinear_adjoint_kernel<true>, dim3(cdiv(nf, 256)), dim3(256), 0, s, (const float *)dcoords,
                           dfield, B, 3, fD, fH, fW, D, H, W, groups, 1, (const int *)fbox);
    }
    return check_launch("warp_volume_bwd");
}

extern "C" size_t mphip_warp_field_compose_bwd_workspace_bytes(int B, int G) {
    if (B <= 0 || G <= 0) return 0;
    const size_t theta = (size_t)B * 3 * cdiv((size_t)G * G * G, TG_CHUNK) * 4 * sizeof(double);
    const size_t axis = (size_t)B * 3 * G * G * G * sizeof(float) * 2;  // two intermediates of the separable adjoint (upper bound)
    return theta + axis;
}

extern "C" int mphip_warp_field_compose_bwd(const float *dw, const float *base_tbl, float *dtheta, float *dem, int B, int eD,
                                            int eH, int eW, int G, void *workspace, size_t workspace_bytes, void *stream) {
    MPHIP_REQUIRE(dw && base_tbl && (dtheta || dem), "warp_field_compose_bwd: null pointer");
    MPHIP_REQUIRE(B > 0 && eD > 0 && eH > 0 && eW > 0 && G > 0, "warp_field_compose_bwd: bad dims");
    hipStream_t s = (hipStream_t)stream;
    if (dtheta) {
        const size_t need = mphip_warp_field_compose_bwd_workspace_bytes(B, G);
        if (!workspace || workspace_bytes < need) {
            set_error("warp_field_compose_bwd: workspace %zu bytes < required %zu", workspace_bytes, need);
            return MPHIP_EWORKSPACE;
        }
        const int chunks = cdiv((size_t)G * G * G, TG_CHUNK);
        hipLaunchKernelGGL(theta_grad_partial_kernel, dim3(B * 3 * chunks), dim3(256), 0, s, dw, base_tbl, (double *)workspace, G,
                           chunks);
        hipLaunchKernelGGL(theta_grad_finalize_kernel, dim3(cdiv(B * 12, 64)), dim3(64), 0, s, (const double *)workspace, dtheta,
                           B * 12, chunks);
    }
    if (dem) {
        const size_t need = mphip_warp_field_compose_bwd_workspace_bytes(B, G);
        if (!workspace || workspace_bytes < need) {
            set_error("warp_field_compose_bwd: workspace %zu bytes < required %zu", workspace_bytes, need);
            return MPHIP_EWORKSPACE;
        }
        // separable: W, then H, then D (each pass gathers <= ~10 outputs per input along one axis)
        const size_t theta_bytes = (size_t)B * 3 * cdiv((size_t)G * G * G, TG_CHUNK) * 4 * sizeof(double);
        float *t1 = (float *)((char *)workspace + theta_bytes);          // [B*3][G][G][eW]
        float *t2 = t1 + (size_t)B * 3 * G * G * eW;                     // [B*3][G][eH][eW]
        const size_t n1 = (size_t)B * 3 * G * G * eW, n2 = (size_t)B * 3 * G * eH * eW, n3 = (size_t)B * 3 * eD * eH * eW;
        hipLaunchKernelGGL(resize_adjoint_axis_kernel<false>, dim3(cdiv(n1, 256)), dim3(256), 0, s, dw, t1, (size_t)B * 3 * G * G, eW, G,
                           (size_t)1);
        hipLaunchKernelGGL(resize_adjoint_axis_kernel<false>, dim3(cdiv(n2, 256)), dim3(256), 0, s, (const float *)t1, t2,
                           (size_t)B * 3 * G, eH, G, (size_t)eW);
        hipLaunchKernelGGL(resize_adjoint_axis_kernel<false>, dim3(cdiv(n3, 256)), dim3(256), 0, s, (const float *)t2, dem,
                           (size_t)B * 3, eD, G, (size_t)eH * eW);
    }
    return check_launch("warp_field_compose_bwd");
}

extern "C" int mphip_rt_theta_bwd(const float *rot, const float *tr, const float *dtheta, float *drot, float *dtr, int B,
                                  int invert, void *stream) {
    MPHIP_REQUIRE(rot && tr && dtheta && drot && dtr, "rt_theta_bwd: null pointer");
    MPHIP_REQUIRE(B > 0, "rt_theta_bwd: bad batch");
    hipLaunchKernelGGL(rt_theta_bwd_kernel, dim3(cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, rot, tr, dtheta, drot, dtr, B,
                       invert);
    return check_launch("rt_theta_bwd");
}

#ifdef MPHIP_K2_TRACE
extern "C" int mphip_debug_k2_trace(unsigned long long *host_out /* 4096 * 4 */) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mphip::g_k2_trace), sizeof(unsigned long long) * 4096 * 4) == hipSuccess ? 0 : -1;
}
#endif
