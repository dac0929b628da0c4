// dmm_graph.hip -- make a captured HIP graph safe to replay on this runtime: memset / 1-D device-to-device memcpy nodes
// become KERNEL nodes.
//
// Why: on the ROCm runtime of this image (torch 2.10 + rocm 7.0, MI355X) a hipMemsetAsync captured into a graph is not
// reliably ordered before the kernel node that follows it on REPLAY (profiles/r04_graph_memset_node.txt: the frame loop's
// count tables; round 6: MIOpen's bf16 weight-gradient solvers clear their fp32 split-K workspace with hipMemsetAsync -- in
// a captured training step the replayed gradients of the 3x3 convolutions came out non-finite, eager launches never did;
// tools/train_encoder_diag.py).  The library's own kernels never use memset nodes (dmm::zero_async); this entry extends
// the same rule to graphs that contain OTHER libraries' launches (MIOpen, hipBLASLt, torch): after the capture and before
// hipGraphInstantiate the graph is walked, every memset node is replaced by a fill kernel node with the same dependencies
// and dependents, optionally every 1-D device-to-device memcpy node by a copy kernel node.  A kernel node is ordered like
// every other kernel of the chain.
//
// Reference: nothing in the reference (it launches eagerly, train.py:296-307); this is plumbing of the HIP-graph replay of
// the trainer's step (dmm_net_amd/train_encoder.py).
#include "dmm_common.h"

#include <cstring>
#include <vector>

namespace dmm {

typedef uint32_t u32x4g __attribute__((ext_vector_type(4)));

// one thread = 16 bytes of one row; byte o of a row holds byte (o % 4) of `pat` (the element pattern repeated from the row's
// first byte: element sizes 1, 2, 4)
__global__ __launch_bounds__(256) void graph_fill_kernel(uint8_t *dst, size_t pitch, size_t width_bytes, uint32_t pat) {
    uint8_t *row = dst + (size_t)blockIdx.y * pitch;
    const size_t o = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (o >= width_bytes) return;
    if (o + 16 <= width_bytes && (reinterpret_cast<uintptr_t>(row + o) & 15) == 0) {
        const u32x4g v = {pat, pat, pat, pat};
        *reinterpret_cast<u32x4g *>(row + o) = v;
        return;
    }
    const size_t end = o + 16 < width_bytes ? o + 16 : width_bytes;
    for (size_t i = o; i < end; ++i) row[i] = (uint8_t)(pat >> (8 * (i & 3)));
}

__global__ __launch_bounds__(256) void graph_copy_kernel(uint8_t *dst, const uint8_t *src, size_t bytes) {
    const size_t o = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (o >= bytes) return;
    if (o + 16 <= bytes && ((reinterpret_cast<uintptr_t>(dst + o) | reinterpret_cast<uintptr_t>(src + o)) & 15) == 0) {
        *reinterpret_cast<u32x4g *>(dst + o) = *reinterpret_cast<const u32x4g *>(src + o);
        return;
    }
    const size_t end = o + 16 < bytes ? o + 16 : bytes;
    for (size_t i = o; i < end; ++i) dst[i] = src[i];
}

static int graph_err(hipError_t e) {
    set_last_hip_error((int)e);
    return DMM_ERR_LAUNCH;
}

static hipError_t replace_node(hipGraph_t graph, hipGraphNode_t old, const hipKernelNodeParams &kp) {
    size_t nd = 0, nt = 0;
    hipError_t e = hipGraphNodeGetDependencies(old, nullptr, &nd);
    if (e != hipSuccess) return e;
    std::vector<hipGraphNode_t> deps(nd ? nd : 1);
    if (nd && (e = hipGraphNodeGetDependencies(old, deps.data(), &nd)) != hipSuccess) return e;
    e = hipGraphNodeGetDependentNodes(old, nullptr, &nt);
    if (e != hipSuccess) return e;
    std::vector<hipGraphNode_t> outs(nt ? nt : 1);
    if (nt && (e = hipGraphNodeGetDependentNodes(old, outs.data(), &nt)) != hipSuccess) return e;
    hipGraphNode_t k = nullptr;
    if ((e = hipGraphAddKernelNode(&k, graph, nd ? deps.data() : nullptr, nd, &kp)) != hipSuccess) return e;
    for (size_t i = 0; i < nt; ++i)
        if ((e = hipGraphAddDependencies(graph, &k, &outs[i], 1)) != hipSuccess) return e;
    return hipGraphDestroyNode(old);
}

}  // namespace dmm

extern "C" int dmm_graph_nodes_to_kernels(void *graph_, int flags, int *n_memset, int *n_memcpy, int *n_left) {
    hipGraph_t graph = (hipGraph_t)graph_;
    if (!graph) return DMM_ERR_BAD_ARG;
    int done_set = 0, done_cpy = 0, left = 0;
    size_t n = 0;
    if (hipGraphGetNodes(graph, nullptr, &n) != hipSuccess) return dmm::graph_err(hipGetLastError());
    std::vector<hipGraphNode_t> nodes(n ? n : 1);
    if (n && hipGraphGetNodes(graph, nodes.data(), &n) != hipSuccess) return dmm::graph_err(hipGetLastError());
    for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType type;
        if (hipGraphNodeGetType(nodes[i], &type) != hipSuccess) return dmm::graph_err(hipGetLastError());
        if (type == hipGraphNodeTypeMemset) {
            hipMemsetParams p;
            if (hipGraphMemsetNodeGetParams(nodes[i], &p) != hipSuccess) return dmm::graph_err(hipGetLastError());
            const bool ok = (flags & 1) && (p.elementSize == 1 || p.elementSize == 2 || p.elementSize == 4);
            if (!ok) { ++left; continue; }
            if (p.width == 0 || p.height == 0) continue;
            uint32_t pat = p.value;
            if (p.elementSize == 1) pat = (pat & 0xffu) * 0x01010101u;
            else if (p.elementSize == 2) pat = (pat & 0xffffu) | (pat << 16);
            uint8_t *dst = (uint8_t *)p.dst;
            size_t pitch = p.height > 1 ? p.pitch : 0, wb = p.width * (size_t)p.elementSize;
            void *args[] = {&dst, &pitch, &wb, &pat};
            hipKernelNodeParams kp = {};
            kp.func = (void *)dmm::graph_fill_kernel;
            kp.blockDim = dim3(256);
            kp.gridDim = dim3((unsigned)((wb + 4095) / 4096), (unsigned)p.height);
            kp.kernelParams = args;
            const hipError_t e = dmm::replace_node(graph, nodes[i], kp);
            if (e != hipSuccess) return dmm::graph_err(e);
            ++done_set;
        } else if (type == hipGraphNodeTypeMemcpy) {
            // (a node captured from a plain hipMemcpyAsync is a 1-D node: this runtime's hipGraphMemcpyNodeGetParams leaves the
            // 3-D description untouched for it and there is no 1-D getter -- such nodes are counted in n_left and stay;
            // no mis-ordering has been observed for memcpy nodes, only for memset nodes)
            hipMemcpy3DParms p;
            memset(&p, 0, sizeof(p));
            if (hipGraphMemcpyNodeGetParams(nodes[i], &p) != hipSuccess) { (void)hipGetLastError(); ++left; continue; }
            if (!p.srcPtr.ptr || !p.dstPtr.ptr || p.extent.width == 0) { ++left; continue; }
            const bool flat = p.extent.height <= 1 && p.extent.depth <= 1 && !p.srcArray && !p.dstArray &&
                              p.srcPos.x == 0 && p.srcPos.y == 0 && p.srcPos.z == 0 && p.dstPos.x == 0 && p.dstPos.y == 0 &&
                              p.dstPos.z == 0;
            bool d2d = p.kind == hipMemcpyDeviceToDevice;
            if (p.kind == hipMemcpyDefault && flat) {         // unified addressing: look at where the two pointers live
                hipPointerAttribute_t as, ad;
                d2d = hipPointerGetAttributes(&as, p.srcPtr.ptr) == hipSuccess && as.type == hipMemoryTypeDevice &&
                      hipPointerGetAttributes(&ad, p.dstPtr.ptr) == hipSuccess && ad.type == hipMemoryTypeDevice;
                (void)hipGetLastError();
            }
            if (!((flags & 2) && flat && d2d)) { ++left; continue; }
            uint8_t *dst = (uint8_t *)p.dstPtr.ptr;
            const uint8_t *src = (const uint8_t *)p.srcPtr.ptr;
            size_t bytes = p.extent.width;
            if (bytes == 0) continue;
            void *args[] = {&dst, &src, &bytes};
            hipKernelNodeParams kp = {};
            kp.func = (void *)dmm::graph_copy_kernel;
            kp.blockDim = dim3(256);
            kp.gridDim = dim3((unsigned)((bytes + 4095) / 4096));
            kp.kernelParams = args;
            const hipError_t e = dmm::replace_node(graph, nodes[i], kp);
            if (e != hipSuccess) return dmm::graph_err(e);
            ++done_cpy;
        }
    }
    if (n_memset) *n_memset = done_set;
    if (n_memcpy) *n_memcpy = done_cpy;
    if (n_left) *n_left = left;
    return DMM_OK;
}
