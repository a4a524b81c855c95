/* The one-call entry and the index-contract ops from plain C: no Python, no torch — device memory from
 * the HIP runtime, pointers + sizes + a stream across include/mphip.h.  The checker is the plain-C oracle
 * (oracle/libmphip_oracle.so, orc_*) and a committed golden of the CPU oracle; this file is test code.
 *
 *   plan_smoke <tests/golden/plan_c_manifest.txt> <tests/golden/plan_c_expected.bin>
 *
 *  A. mphip_warp_volume (K2): coordinates, floor indices AND values bit-exact vs orc_warp_coords / orc_grid_sample3d;
 *     mphip_warp_volume_dsum (K3) vs the oracle's depth sum.
 *  B. a 3x3x3 conv on the f16x3 kernel with a CALLER-BUILT range descriptor {0, 0, bound, 0} vs orc_conv3d (double accumulate).
 *  C. mphip_hot_slice_plan_create / _forward / _destroy on the manifest's integer-PRNG state-dict (206 tensors, regenerated
 *     here with the same 64-bit LCG) at a 16x16x16 volume: max-abs vs the committed oracle output < 1e-3; workspace contract.
 * Built and run by tests/test_gpu_plan.py::test_plan_from_plain_c (needs a GPU).                                        */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "mphip.h"
#include "../../megaportrait-hack_amd/csrc/linspace_tables.h" /* data: the captured torch.linspace bit patterns */

/* oracle/hotpath_c.c */
void orc_resize_trilinear(const float *in, float *out, int NC, int iD, int iH, int iW, int oD, int oH, int oW, int align);
void orc_warp_coords(const float *field, const float *lin_d, const float *lin_h, const float *lin_w, int B, int D, int H, int W,
                     float *coords, int32_t *idx);
void orc_grid_sample3d(const float *v, const float *coords, int B, int C, int D, int H, int W, float *out, int dsum);
void orc_conv3d(const float *in, const float *wt, const float *bias, float *out, int N, int Ci, int Co, int D, int H, int W, int k);

#define CHECK_HIP(x)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (x);                                                                             \
        if (e_ != hipSuccess) {                                                                          \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);       \
            return 2;                                                                                    \
        }                                                                                                \
    } while (0)
#define CHECK_MP(x)                                                                                      \
    do {                                                                                                 \
        int rc_ = (x);                                                                                   \
        if (rc_ != MPHIP_OK) {                                                                           \
            fprintf(stderr, "%s -> %d: %s (%s:%d)\n", #x, rc_, mphip_last_error(), __FILE__, __LINE__);  \
            return 1;                                                                                    \
        }                                                                                                \
    } while (0)

/* oracle/hotpath_ref.py:_lcg_uniform — 64-bit LCG, top 24 bits -> U[-1,1) fp32, then * scale + shift in fp32 */
static void lcg_fill(float *dst, size_t n, uint64_t seed, float scale, float shift) {
    const uint64_t a = 6364136223846793005ull, c = 1442695040888963407ull;
    uint64_t st = seed * 2654435761ull + 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) {
        st = a * st + c;
        const float u = (float)((double)(st >> 40) / 16777216.0 * 2.0 - 1.0);
        volatile float m = u * scale; /* one rounding per op, like the tensor expression `t * scale + shift` */
        dst[i] = m + shift;
    }
}

static float *to_device(const float *h, size_t n) {
    float *d = NULL;
    if (hipMalloc((void **)&d, n * 4) != hipSuccess) return NULL;
    if (hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice) != hipSuccess) return NULL;
    return d;
}

static float bits_to_float(uint32_t b) {
    float f;
    memcpy(&f, &b, 4);
    return f;
}

static int part_a_warps(hipStream_t stream) {
    enum { B = 2, C = 16, D = 16, H = 16, W = 16, S = D * H * W };
    const size_t nv = (size_t)B * C * S, nf = (size_t)B * 3 * S, nc = (size_t)B * S * 3;
    float *v = malloc(nv * 4), *f = malloc(nf * 4), *lin = malloc(16 * 4);
    float *coords = malloc(nc * 4), *want = malloc(nv * 4), *want_ds = malloc((size_t)B * C * H * W * 4);
    int32_t *idx = malloc(nc * 4);
    for (int i = 0; i < 16; ++i) lin[i] = bits_to_float(MPHIP_TBL_LINSPACE_16[i]);
    lcg_fill(v, nv, 101, 1.7f, 0.0f);
    lcg_fill(f, nf, 102, 0.9f, 0.5f); /* field in (-0.4, 1.4): samples stay near the low corner, some clipped at 0 (the reference's quirk) */
    orc_warp_coords(f, lin, lin, lin, B, D, H, W, coords, idx); /* field already at (D,H,W): the align_corners=True resize is the identity */
    orc_grid_sample3d(v, coords, B, C, D, H, W, want, 0);
    orc_grid_sample3d(v, coords, B, C, D, H, W, want_ds, 1);
    float *dv = to_device(v, nv), *df = to_device(f, nf), *dlin = to_device(lin, 16), *dout, *dcoords, *dds;
    int32_t *didx;
    void *ws;
    if (!dv || !df || !dlin) return 2;
    CHECK_HIP(hipMalloc((void **)&dout, nv * 4));
    CHECK_HIP(hipMalloc((void **)&dcoords, nc * 4));
    CHECK_HIP(hipMalloc((void **)&didx, nc * 4));
    CHECK_HIP(hipMalloc((void **)&dds, (size_t)B * C * H * W * 4));
    const size_t wsb = mphip_warp_workspace_bytes(B, D, H, W);
    CHECK_HIP(hipMalloc(&ws, wsb));
    CHECK_MP(mphip_warp_volume(dv, df, dlin, dlin, dlin, dout, dcoords, didx, NULL, B, C, D, H, W, D, H, W, ws, wsb, stream));
    CHECK_MP(mphip_warp_volume_dsum(dv, df, dlin, dlin, dlin, dds, B, C, D, H, W, D, H, W, ws, wsb, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    float *got = malloc(nv * 4), *gc = malloc(nc * 4), *gds = malloc((size_t)B * C * H * W * 4);
    int32_t *gi = malloc(nc * 4);
    CHECK_HIP(hipMemcpy(got, dout, nv * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(gc, dcoords, nc * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(gi, didx, nc * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(gds, dds, (size_t)B * C * H * W * 4, hipMemcpyDeviceToHost));
    if (memcmp(gc, coords, nc * 4) != 0) { fprintf(stderr, "K2 coordinates differ from the oracle bitwise\n"); return 1; }
    if (memcmp(gi, idx, nc * 4) != 0) { fprintf(stderr, "K2 floor indices differ from the oracle\n"); return 1; }
    if (memcmp(got, want, nv * 4) != 0) { fprintf(stderr, "K2 values differ from the oracle bitwise\n"); return 1; }
    double worst = 0.0, scale = 0.0;
    for (size_t i = 0; i < (size_t)B * C * H * W; ++i) {
        worst = fmax(worst, fabs((double)gds[i] - want_ds[i]));
        scale = fmax(scale, fabs((double)want_ds[i]));
    }
    printf("K2: coords / indices / values bit-exact vs the C oracle (%zu voxels x %d channels); K3 depth sum max-abs %.3e (|.|max %.2f)\n",
           (size_t)B * S, C, worst, scale);
    if (!(worst <= 1e-5 * scale)) return 1;
    return 0;
}

static int part_b_conv(hipStream_t stream) {
    enum { N = 2, Ci = 32, Co = 96, D = 4, H = 8, W = 8, S = D * H * W };
    const size_t nx = (size_t)N * Ci * S, ny = (size_t)N * Co * S, nw = (size_t)Co * Ci * 27;
    float *x = malloc(nx * 4), *w = malloc(nw * 4), *b = malloc(Co * 4), *want = malloc(ny * 4), *got = malloc(ny * 4);
    lcg_fill(x, nx, 201, 40.0f, 3.0f); /* un-normalised activations: |x| up to 43 */
    lcg_fill(w, nw, 202, 0.05f, 0.0f);
    lcg_fill(b, Co, 203, 0.5f, 0.0f);
    orc_conv3d(x, w, b, want, N, Ci, Co, D, H, W, 3);
    if (!mphip_conv3d_supported(N, Ci, Co, D, H, W, 3, 1)) { fprintf(stderr, "f16x3 not supported for the test shape?\n"); return 1; }
    float *dx = to_device(x, nx), *dw = to_device(w, nw), *db = to_device(b, Co), *dy;
    /* caller-built range descriptor: {scale = 0 (derive), 1/scale, bound on max|x|, n partials = 0} */
    float bound = 0.0f;
    for (size_t i = 0; i < nx; ++i) bound = fmaxf(bound, fabsf(x[i]));
    float rd[4] = {0.0f, 0.0f, bound, 0.0f};
    float *drange = to_device(rd, 4);
    void *dwp, *ws = NULL;
    if (!dx || !dw || !db || !drange) return 2;
    CHECK_HIP(hipMalloc((void **)&dy, ny * 4));
    const size_t pb = mphip_packed_weight_bytes(Co, Ci, 3, 1);
    CHECK_HIP(hipMalloc(&dwp, pb));
    CHECK_MP(mphip_pack_conv_weight(dw, dwp, Co, Ci, 3, 1, stream));
    const size_t wsb = mphip_conv3d_workspace_bytes(N, Ci, Co, D, H, W, 3, 1);
    if (wsb) CHECK_HIP(hipMalloc(&ws, wsb));
    CHECK_MP(mphip_conv3d_fwd(dx, drange, dwp, db, dy, N, Ci, Co, D, H, W, 3, 1, ws, wsb, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    CHECK_HIP(hipMemcpy(got, dy, ny * 4, hipMemcpyDeviceToHost));
    double worst = 0.0, scale = 0.0;
    for (size_t i = 0; i < ny; ++i) {
        worst = fmax(worst, fabs((double)got[i] - want[i]));
        scale = fmax(scale, fabs((double)want[i]));
    }
    unsigned long long sat = 0;
    CHECK_MP(mphip_f16x3_saturation_count(&sat, 1));
    printf("conv3d 3x3x3 f16x3, caller-built range descriptor (bound %.2f): max-abs %.3e on |y|max %.2f, saturated operands %llu\n", bound,
           worst, scale, sat);
    if (!(worst <= 2e-5 * scale) || sat != 0) return 1;
    return 0;
}

static int part_c_plan(const char *manifest, const char *expected, hipStream_t stream) {
    enum { B = 1, C = 96, D = 16, H = 16, W = 16, MAXT = 256 };
    FILE *fm = fopen(manifest, "r");
    if (!fm) { fprintf(stderr, "cannot open %s\n", manifest); return 2; }
    static char names[MAXT][128];
    const char *name_ptrs[MAXT];
    const void *tensor_ptrs[MAXT];
    float *inputs[8] = {0};
    static const char *input_names[8] = {"input.vs", "input.es", "input.Rs", "input.ts", "input.zs", "input.Rd", "input.td", "input.zd"};
    int nt = 0;
    char line[512];
    while (fgets(line, sizeof line, fm)) {
        if (line[0] == '#' || line[0] == '\n') continue;
        char nm[128];
        unsigned long long n, seed;
        unsigned sb, hb;
        if (sscanf(line, "%127s %llu %llu %x %x", nm, &n, &seed, &sb, &hb) != 5) { fprintf(stderr, "bad manifest line: %s", line); return 2; }
        float *h = malloc((size_t)n * 4);
        lcg_fill(h, (size_t)n, seed, bits_to_float(sb), bits_to_float(hb));
        float *d = to_device(h, (size_t)n);
        free(h);
        if (!d) return 2;
        int is_input = 0;
        for (int i = 0; i < 8; ++i)
            if (strcmp(nm, input_names[i]) == 0) { inputs[i] = d; is_input = 1; }
        if (!is_input) {
            if (nt >= MAXT) { fprintf(stderr, "too many tensors\n"); return 2; }
            strcpy(names[nt], nm);
            name_ptrs[nt] = names[nt];
            tensor_ptrs[nt] = d;
            ++nt;
        }
    }
    fclose(fm);
    for (int i = 0; i < 8; ++i)
        if (!inputs[i]) { fprintf(stderr, "manifest lacks %s\n", input_names[i]); return 2; }
    mphip_hot_slice_plan *plan = NULL;
    /* a missing tensor is reported by name, not a crash */
    int rc = mphip_hot_slice_plan_create(name_ptrs, tensor_ptrs, nt - 1, C, D, H, W, 0, &plan);
    if (rc != MPHIP_EINVAL || !strstr(mphip_last_error(), "missing")) { fprintf(stderr, "expected 'missing' error, got %d: %s\n", rc, mphip_last_error()); return 1; }
    CHECK_MP(mphip_hot_slice_plan_create(name_ptrs, tensor_ptrs, nt, C, D, H, W, 0, &plan));
    const size_t wsb = mphip_hot_slice_workspace_bytes(plan, B);
    if (wsb == 0) { fprintf(stderr, "workspace query returned 0\n"); return 1; }
    void *ws;
    float *dout;
    const size_t no = (size_t)B * C * H * W;
    CHECK_HIP(hipMalloc(&ws, wsb));
    CHECK_HIP(hipMalloc((void **)&dout, no * 4));
    rc = mphip_hot_slice_forward(plan, inputs[0], inputs[1], inputs[2], inputs[3], inputs[4], inputs[5], inputs[6], inputs[7], dout, B, ws, wsb / 2,
                                 stream);
    if (rc != MPHIP_EWORKSPACE) { fprintf(stderr, "expected MPHIP_EWORKSPACE for half the workspace, got %d\n", rc); return 1; }
    float *got = malloc(no * 4), *got2 = malloc(no * 4), *want = malloc(no * 4);
    for (int rep = 0; rep < 2; ++rep) { /* second call: packs cached, workspace reused — same bits */
        CHECK_MP(mphip_hot_slice_forward(plan, inputs[0], inputs[1], inputs[2], inputs[3], inputs[4], inputs[5], inputs[6], inputs[7], dout, B, ws,
                                         wsb, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        CHECK_HIP(hipMemcpy(rep ? got2 : got, dout, no * 4, hipMemcpyDeviceToHost));
    }
    if (memcmp(got, got2, no * 4) != 0) { fprintf(stderr, "two forwards of the same plan differ\n"); return 1; }
    FILE *fe = fopen(expected, "rb");
    if (!fe || fread(want, 4, no, fe) != no) { fprintf(stderr, "cannot read %zu floats from %s\n", no, expected); return 2; }
    fclose(fe);
    double worst = 0.0, scale = 0.0;
    for (size_t i = 0; i < no; ++i) {
        if (!(got[i] == got[i])) { fprintf(stderr, "NaN in the output\n"); return 1; }
        worst = fmax(worst, fabs((double)got[i] - want[i]));
        scale = fmax(scale, fabs((double)want[i]));
    }
    printf("hot slice plan (%d state-dict tensors, volume %dx%dx%dx%d, workspace %.1f MB): max-abs vs the oracle golden %.3e (|out|max %.2f)\n", nt,
           C, D, H, W, wsb / 1e6, worst, scale);
    mphip_hot_slice_plan_destroy(plan);
    if (!(worst < 1e-3)) return 1;
    return 0;
}

int main(int argc, char **argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s manifest.txt expected.bin\n", argv[0]); return 2; }
    if (mphip_version() != MPHIP_ABI_VERSION) { fprintf(stderr, "library ABI %d != header ABI %d\n", mphip_version(), MPHIP_ABI_VERSION); return 1; }
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    int rc = part_a_warps(stream);
    if (rc) return rc;
    rc = part_b_conv(stream);
    if (rc) return rc;
    rc = part_c_plan(argv[1], argv[2], stream);
    if (rc) return rc;
    printf("PLAN C ABI OK\n");
    return 0;
}
