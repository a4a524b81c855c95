// Library-level entry points: version and the thread-local error string.
#include <string.h>

#include <algorithm>
#include <vector>

#include "mphip_common.h"

namespace mphip {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace mphip

namespace mphip {
// conv arithmetic policy of the calling thread (mphip_conv3d_set_half_products): a launch reads it when it is issued
static thread_local int g_half_products = 0;
bool conv_half_products() { return g_half_products != 0; }
}  // namespace mphip

extern "C" int mphip_conv3d_set_half_products(int enable) {
    const int prev = mphip::g_half_products;
    mphip::g_half_products = enable ? 1 : 0;
    return prev;
}

extern "C" int mphip_version(void) { return MPHIP_ABI_VERSION; }
// bit 0: some translation unit of this library was compiled with a timing-only ablation on (mphip_ablate.h): its results are wrong by design
extern "C" __attribute__((weak)) int mphip_ablated_build_marker;
extern "C" int mphip_build_flags(void) { return &mphip_ablated_build_marker != nullptr ? 1 : 0; }
extern "C" const char *mphip_last_error(void) { return mphip::g_err; }

namespace mphip {
__global__ void __launch_bounds__(256) zero_fill_kernel(float4 *__restrict__ p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
}  // namespace mphip

// ---- hipGraph hygiene: MEMSET nodes -> kernel nodes ------------------------------------------------------------------------------
// On ROCm 7.x a MEMSET node of a captured graph is not reliably ordered with the kernel nodes around it, even in a strictly linear
// chain (r03: stale f16x3 pack headers in ~40 % of replays until the library's own hipMemsetAsync calls became kernels; r05: the
// "flaky graphed loss" of r04 — ATen's multi-block reduction zeroes its semaphores with hipMemsetAsync, the last block to finish writes
// the result, and with a semaphore that still holds the previous replay's count NO block is the last: the loss tensor keeps its old
// value while the step itself is right; tools/dbg_replay_stale_loss.py, tools/dbg_graph_topology.py: 262 nodes, 261 edges, ONE memset).
// A captured graph may contain memsets the library never issued (any torch op in the user's loss).  This entry rewrites a hipGraph_t
// in place: every 1-D MEMSET node becomes a kernel node (same bytes, same predecessors and successors).  Call it between capture and
// instantiation (torch.cuda.CUDAGraph(keep_graph=True) -> raw_cuda_graph() -> this -> instantiate()).
namespace mphip {
__global__ void __launch_bounds__(256) graph_memset_kernel(unsigned char *__restrict__ dst, unsigned value, unsigned elem, size_t count) {
    // count elements of `elem` bytes (1, 2 or 4), each set to the low bytes of `value` (hipMemsetParams semantics)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        if (elem == 4) reinterpret_cast<unsigned *>(dst)[i] = value;
        else if (elem == 2) reinterpret_cast<unsigned short *>(dst)[i] = (unsigned short)value;
        else dst[i] = (unsigned char)value;
    }
}
}  // namespace mphip

extern "C" int mphip_graph_memsets_to_kernels(void *graph_, int *replaced_out) {
    MPHIP_REQUIRE(graph_, "graph_memsets_to_kernels: null graph");
    hipGraph_t graph = (hipGraph_t)graph_;
    size_t n = 0;
    if (hipGraphGetNodes(graph, nullptr, &n) != hipSuccess) { mphip::set_error("graph_memsets_to_kernels: hipGraphGetNodes failed"); return MPHIP_ELAUNCH; }
    std::vector<hipGraphNode_t> nodes(n);
    if (n && hipGraphGetNodes(graph, nodes.data(), &n) != hipSuccess) { mphip::set_error("graph_memsets_to_kernels: hipGraphGetNodes failed"); return MPHIP_ELAUNCH; }
    int replaced = 0;
    // (kernel parameters are copied by hipGraphAddKernelNode; these live only across the call)
    for (hipGraphNode_t node : nodes) {
        hipGraphNodeType type;
        if (hipGraphNodeGetType(node, &type) != hipSuccess || type != hipGraphNodeTypeMemset) continue;
        hipMemsetParams mp;
        if (hipGraphMemsetNodeGetParams(node, &mp) != hipSuccess) { mphip::set_error("graph_memsets_to_kernels: hipGraphMemsetNodeGetParams failed"); return MPHIP_ELAUNCH; }
        if (mp.height > 1 || (mp.elementSize != 1 && mp.elementSize != 2 && mp.elementSize != 4)) continue;   // (2-D memsets: left alone)
        size_t np = 0, ns = 0;
        if (hipGraphNodeGetDependencies(node, nullptr, &np) != hipSuccess || hipGraphNodeGetDependentNodes(node, nullptr, &ns) != hipSuccess) {
            mphip::set_error("graph_memsets_to_kernels: dependency counts");
            return MPHIP_ELAUNCH;
        }
        std::vector<hipGraphNode_t> preds(np), succs(ns);
        if (np && hipGraphNodeGetDependencies(node, preds.data(), &np) != hipSuccess) { mphip::set_error("graph_memsets_to_kernels: dependencies"); return MPHIP_ELAUNCH; }
        if (ns && hipGraphNodeGetDependentNodes(node, succs.data(), &ns) != hipSuccess) { mphip::set_error("graph_memsets_to_kernels: dependents"); return MPHIP_ELAUNCH; }
        unsigned char *dst = (unsigned char *)mp.dst;
        unsigned value = mp.value, elem = mp.elementSize;
        size_t count = mp.width;
        void *args[4] = {&dst, &value, &elem, &count};
        hipKernelNodeParams kp;
        memset(&kp, 0, sizeof(kp));
        kp.func = (void *)mphip::graph_memset_kernel;
        kp.blockDim = dim3(256);
        kp.gridDim = dim3((unsigned)std::min<size_t>(1024, (count + 255) / 256 ? (count + 255) / 256 : 1));
        kp.kernelParams = args;
        hipGraphNode_t knode;
        if (hipGraphAddKernelNode(&knode, graph, preds.data(), np, &kp) != hipSuccess) {
            mphip::set_error("graph_memsets_to_kernels: hipGraphAddKernelNode failed (%s)", hipGetErrorString(hipGetLastError()));
            return MPHIP_ELAUNCH;
        }
        for (hipGraphNode_t sct : succs)
            if (hipGraphAddDependencies(graph, &knode, &sct, 1) != hipSuccess) { mphip::set_error("graph_memsets_to_kernels: hipGraphAddDependencies failed"); return MPHIP_ELAUNCH; }
        if (hipGraphDestroyNode(node) != hipSuccess) { mphip::set_error("graph_memsets_to_kernels: hipGraphDestroyNode failed"); return MPHIP_ELAUNCH; }
        ++replaced;
    }
    if (replaced_out) *replaced_out = replaced;
    return MPHIP_OK;
}

// Counts the MEMSET nodes `graph` (and its child graphs) still holds: what mphip_graph_memsets_to_kernels left alone.
static int count_memset_nodes(hipGraph_t graph, int depth, int *count) {
    size_t n = 0;
    if (hipGraphGetNodes(graph, nullptr, &n) != hipSuccess) return MPHIP_ELAUNCH;
    std::vector<hipGraphNode_t> nodes(n);
    if (n && hipGraphGetNodes(graph, nodes.data(), &n) != hipSuccess) return MPHIP_ELAUNCH;
    for (hipGraphNode_t node : nodes) {
        hipGraphNodeType type;
        if (hipGraphNodeGetType(node, &type) != hipSuccess) return MPHIP_ELAUNCH;
        if (type == hipGraphNodeTypeMemset) ++*count;
        if (type == hipGraphNodeTypeGraph && depth < 8) {
            hipGraph_t child;
            if (hipGraphChildGraphNodeGetGraph(node, &child) != hipSuccess) return MPHIP_ELAUNCH;
            const int rc = count_memset_nodes(child, depth + 1, count);
            if (rc != MPHIP_OK) return rc;
        }
    }
    return MPHIP_OK;
}

extern "C" int mphip_graph_memset_nodes_left(void *graph_, int *left) {
    MPHIP_REQUIRE(graph_ && left, "graph_memset_nodes_left: null graph or null result");
    *left = 0;
    const int rc = count_memset_nodes((hipGraph_t)graph_, 0, left);
    if (rc != MPHIP_OK) mphip::set_error("graph_memset_nodes_left: graph query failed (%s)", hipGetErrorString(hipGetLastError()));
    return rc;
}

namespace mphip {
__global__ void __launch_bounds__(256) absmax_range_kernel(const float *__restrict__ x, size_t n, float *__restrict__ range) {
    unsigned m = 0;
    const size_t n4 = n / 4;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = x4[i];
        m = max(max(m, range_bits(v.x)), max(range_bits(v.y), max(range_bits(v.z), range_bits(v.w))));
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) m = max(m, range_bits(x[i]));
    range_note_block(m, range, blockIdx.x, gridDim.x);
}

int absmax_range_launch(const float *x, size_t n, float *range, hipStream_t s) {
    const size_t per_block = 256 * 4 * 8;  // ~8 float4 per thread
    const unsigned blocks = (unsigned)std::min<size_t>(2048, (n + per_block - 1) / per_block);
    hipLaunchKernelGGL(absmax_range_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, x, n, range);
    return check_launch("absmax_range");
}
}  // namespace mphip

extern "C" int mphip_absmax_range(const float *x, size_t n, float *range, void *stream) {
    MPHIP_REQUIRE(x && range && n > 0, "absmax_range: null pointer or empty tensor");
    MPHIP_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)range & 3) == 0, "absmax_range: x must be 16-byte aligned");
    return mphip::absmax_range_launch(x, n, range, (hipStream_t)stream);
}
