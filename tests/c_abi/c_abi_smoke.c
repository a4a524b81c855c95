/* Drives libmphip.so from plain C through include/mphip.h only — no Python, no torch: device memory from the HIP
 * runtime, pointers + sizes + a stream across the boundary.  Checks a 1x1x1 conv and an average pool against values
 * computed on the host in this file, the error convention (negative code + message, no crash) and the workspace
 * contract.  Built and run by tests/test_gpu_parity.py::test_c_abi_from_plain_c (needs a GPU).
 *   hipcc -x c ... is not needed: this is C99 compiled by gcc, linked against libamdhip64 and libmphip.            */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "mphip.h"

#define CHECK_HIP(x)                                                                 \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            return 2;                                                                \
        }                                                                            \
    } while (0)

static float lcg(unsigned *s) {
    *s = *s * 1664525u + 1013904223u;
    return (float)((*s >> 8) & 0xFFFF) / 32768.0f - 1.0f;
}

int main(void) {
    if (mphip_version() <= 0) { fprintf(stderr, "bad version\n"); return 1; }
    const int N = 2, Ci = 40, Co = 24, D = 2, H = 4, W = 8, S = D * H * W;
    const size_t nx = (size_t)N * Ci * S, ny = (size_t)N * Co * S, nw = (size_t)Co * Ci;
    float *hx = malloc(nx * 4), *hw = malloc(nw * 4), *hb = malloc(Co * 4), *hy = malloc(ny * 4), *ref = malloc(ny * 4);
    unsigned seed = 12345u;
    for (size_t i = 0; i < nx; ++i) hx[i] = lcg(&seed);
    for (size_t i = 0; i < nw; ++i) hw[i] = 0.2f * lcg(&seed);
    for (int i = 0; i < Co; ++i) hb[i] = lcg(&seed);
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Co; ++co)
            for (int v = 0; v < S; ++v) {
                double a = hb[co];
                for (int ci = 0; ci < Ci; ++ci) a += (double)hw[co * Ci + ci] * hx[((size_t)n * Ci + ci) * S + v];
                ref[((size_t)n * Co + co) * S + v] = (float)a;
            }
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    float *dx, *dw, *db, *dy;
    void *dwp, *dws = NULL;
    CHECK_HIP(hipMalloc((void **)&dx, nx * 4));
    CHECK_HIP(hipMalloc((void **)&dw, nw * 4));
    CHECK_HIP(hipMalloc((void **)&db, Co * 4));
    CHECK_HIP(hipMalloc((void **)&dy, ny * 4));
    CHECK_HIP(hipMemcpy(dx, hx, nx * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dw, hw, nw * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(db, hb, Co * 4, hipMemcpyHostToDevice));
    /* 1x1x1 conv, exact fp32 path (precision 0) */
    if (!mphip_conv3d_supported(N, Ci, Co, D, H, W, 1, 0)) { fprintf(stderr, "conv not supported?\n"); return 1; }
    size_t pbytes = mphip_packed_weight_bytes(Co, Ci, 1, 0);
    CHECK_HIP(hipMalloc(&dwp, pbytes));
    if (mphip_pack_conv_weight(dw, dwp, Co, Ci, 1, 0, stream)) { fprintf(stderr, "pack: %s\n", mphip_last_error()); return 1; }
    size_t wbytes = mphip_conv3d_workspace_bytes(N, Ci, Co, D, H, W, 1, 0);
    if (wbytes) CHECK_HIP(hipMalloc(&dws, wbytes));
    if (mphip_conv3d_fwd(dx, NULL, dwp, db, dy, N, Ci, Co, D, H, W, 1, 0, dws, wbytes, stream)) {
        fprintf(stderr, "conv: %s\n", mphip_last_error());
        return 1;
    }
    CHECK_HIP(hipStreamSynchronize(stream));
    CHECK_HIP(hipMemcpy(hy, dy, ny * 4, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (size_t i = 0; i < ny; ++i) worst = fmax(worst, fabs((double)hy[i] - ref[i]));
    printf("conv3d k=1 max-abs error vs host reference: %.3e\n", worst);
    if (!(worst < 1e-4)) return 1;
    /* avg pool 2x2x2 of the conv output: bit-exact sums of eight floats in a fixed order are not promised, 1e-6 is */
    float *dp, *hp = malloc(ny / 8 * 4);
    CHECK_HIP(hipMalloc((void **)&dp, ny / 8 * 4));
    if (mphip_avgpool2(dy, dp, N * Co, D, H, W, stream)) { fprintf(stderr, "pool: %s\n", mphip_last_error()); return 1; }
    CHECK_HIP(hipStreamSynchronize(stream));
    CHECK_HIP(hipMemcpy(hp, dp, ny / 8 * 4, hipMemcpyDeviceToHost));
    worst = 0.0;
    for (int p = 0; p < N * Co; ++p)
        for (int d = 0; d < D / 2; ++d)
            for (int h = 0; h < H / 2; ++h)
                for (int w = 0; w < W / 2; ++w) {
                    double a = 0.0;
                    for (int k = 0; k < 8; ++k)
                        a += hy[(size_t)p * S + (2 * d + (k >> 2)) * H * W + (2 * h + ((k >> 1) & 1)) * W + 2 * w + (k & 1)];
                    worst = fmax(worst, fabs(a / 8.0 - hp[(((size_t)p * (D / 2) + d) * (H / 2) + h) * (W / 2) + w]));
                }
    printf("avgpool2 max-abs error vs host reference: %.3e\n", worst);
    if (!(worst < 1e-6)) return 1;
    /* error convention: negative code, message, no crash, nothing launched */
    int rc = mphip_conv3d_fwd(dx, NULL, dwp, db, dy, N, Ci, Co, D, H, W, 2, 0, dws, wbytes, stream);
    if (rc != MPHIP_EINVAL || strlen(mphip_last_error()) == 0) { fprintf(stderr, "expected MPHIP_EINVAL, got %d\n", rc); return 1; }
    rc = mphip_groupnorm_stats(dy, (float *)dp, N, Co, S, 4, 1e-5f, NULL, 0, stream);
    if (rc != MPHIP_EWORKSPACE) { fprintf(stderr, "expected MPHIP_EWORKSPACE, got %d (%s)\n", rc, mphip_last_error()); return 1; }
    printf("error codes: EINVAL '%s'\n", mphip_last_error());
    printf("C ABI OK\n");
    return 0;
}
