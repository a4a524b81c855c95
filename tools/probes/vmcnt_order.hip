// Does vmcnt retire a younger LOAD before older STORES?  Every wave issues S stores (16 B/lane, a chip-wide burst to fresh lines) and then
// ONE load of an L2-resident line, then waits vmcnt(S) [only the load would have to be done if counters retire out of order] and stamps, then
// vmcnt(0) and stamps.  Reports mean cycles to each stamp and whether the loaded value was correct at the first stamp.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
constexpr int S = 12;
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) probe(f4 *big, const float *small_, unsigned long long *out, int *bad) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const f4 v = {1.f, 2.f, 3.f, (float)tid};
    float ld;
    unsigned long long t0, t1, t2;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll
    for (int i = 0; i < S; ++i) {
        f4 *p = big + (size_t)i * gridDim.x * 256 + tid;
        asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    }
    const float *q = small_ + (threadIdx.x & 63);
    asm volatile("global_load_dword %0, %1, off" : "=v"(ld) : "v"(q) : "memory");
    asm volatile("s_waitcnt vmcnt(%1)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "n"(S) : "memory");
    const float seen = ld;   // (read right after the counted wait)
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
    if (seen != (float)(threadIdx.x & 63) + 0.5f) atomicAdd(bad, 1);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], t1 - t0); atomicAdd(&out[1], t2 - t0); atomicAdd(&out[2], 1ull); }
}
int main() {
    const int blocks = 256 * 8;
    f4 *big; float *sm; unsigned long long *out; int *bad;
    hipMalloc(&big, (size_t)S * blocks * 256 * 16); hipMalloc(&sm, 256); hipMalloc(&out, 24); hipMalloc(&bad, 4);
    std::vector<float> h(64); for (int i = 0; i < 64; ++i) h[i] = i + 0.5f;
    hipMemcpy(sm, h.data(), 256, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(out, 0, 24); hipMemset(bad, 0, 4);
        probe<<<blocks, 256>>>(big, sm, out, bad);
        hipDeviceSynchronize();
        unsigned long long o[3]; int b; hipMemcpy(o, out, 24, hipMemcpyDeviceToHost); hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost);
        printf("rep %d: %d stores then 1 L2-hit load: cycles to vmcnt(%d) %.0f, to vmcnt(0) %.0f; lanes whose load had not landed at vmcnt(%d): %d\n", rep, S, S,
               (double)o[0] / o[2], (double)o[1] / o[2], S, b);
    }
    return 0;
}
