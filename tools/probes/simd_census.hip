// which SIMD does wave w of a 512-thread workgroup land on?  (HW_REG_HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh[12] se[15:13])
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(512) census(unsigned *out) {
    __shared__ unsigned char big[150000];
    big[threadIdx.x] = 1;
    unsigned id = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, offset 0, size 32
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
    __syncthreads();
    if (big[threadIdx.x ^ 1] == 7) out[0] = 0;
}
int main() {
    unsigned *d; hipMalloc(&d, 1024 * 8 * 4);
    census<<<1024, 512>>>(d);
    hipDeviceSynchronize();
    static unsigned h[1024 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int same = 0, diff = 0; int hist[4][8] = {};
    for (int b = 0; b < 1024; ++b) {
        for (int w = 0; w < 4; ++w) { if (((h[b*8+w] >> 4) & 3) == ((h[b*8+w+4] >> 4) & 3)) ++same; else ++diff; }
        for (int w = 0; w < 8; ++w) hist[(h[b*8+w] >> 4) & 3][w]++;
    }
    printf("pairs (w, w+4) on the same SIMD: %d, on different SIMDs: %d\n", same, diff);
    for (int s = 0; s < 4; ++s) { printf("simd %d:", s); for (int w = 0; w < 8; ++w) printf(" w%d=%d", w, hist[s][w]); printf("\n"); }
    for (int b = 0; b < 4; ++b) { printf("block %d simd ids:", b); for (int w = 0; w < 8; ++w) printf(" %u", (h[b*8+w] >> 4) & 3); printf("\n"); }
    return 0;
}
