// dmm_gemm.hip -- the 1x1 convolutions of the inference encoder as ONE library GEMM each, epilogue included.
//
// Reference: the encoder the matching path is fed by -- torchvision bottlenecks (dmm/modules/vision.py:6-38) and the
// conv -> BatchNorm -> ReLU heads (dmm/modules/base.py:35-54, model_encoder.py:136-146).  In channels-last storage a 1x1
// convolution IS the matrix product  Y[rows, Cout] = X[rows, Cin] . W[Cin, Cout]  of the activation matrix, and what
// follows it in a bottleneck -- folded-BatchNorm bias, the residual add, the ReLU -- is an elementwise epilogue of that
// product.  encoder.FastEncoder ran the product through torch (hipBLASLt, bias + ReLU epilogue) but the RESIDUAL form
// through torch.mm + a separate bias / residual / ReLU pass (dmm_bias_act_bf16): three more passes over the widest
// activations of the network (16 launches, 9 % of a ResNet-50 forward at 16 x 255 x 448).  hipBLASLt takes the residual
// as the C operand (beta = 1) next to the bias + ReLU epilogue, so the whole tail is one launch:
//     Y = relu?( X . W + bias (+ residual) ),   fp32 accumulation, fp32 bias, one rounding to bf16.
// MFMA work goes to the library (north_star: matrix cores for the backbone only, through rocm libraries); this file is
// the descriptor plumbing: row-major operands are handed over as the transposed column-major problem
//     Y^T [Cout, rows] = W^T [Cout, Cin] . X^T [Cin, rows]   (no data movement: a row-major [r, c] IS a column-major [c, r]).
// The algorithm is picked per shape ONCE and cached: the library heuristic's first pick.  Opt-in (option DMM_OPT_GEMM_TUNE = n, n <= 24):
// the first call of a shape outside a stream capture times the heuristic's first n candidates on the call's own operands
// and keeps the fastest.  Measured on the config-3 encoder: 0.945 -> 0.937 ms per forward, i.e. the first pick is already
// good at these sizes (rows x Cin x Cout around 2048 x 1024 x 256, 8-12 us per product), so it is off by default.  The
// call is stream-ordered and graph-capturable (a shape first seen DURING a capture is never timed).
#include <hipblaslt/hipblaslt.h>

#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "dmm_common.h"

namespace dmm {
namespace {

struct GemmPlan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t a = nullptr, b = nullptr, c = nullptr, d = nullptr;
    hipblasLtMatmulAlgo_t algo;
    size_t workspace = 0;
    bool ok = false;
    // the heuristic's other candidates until the first un-captured call has timed them
    std::vector<hipblasLtMatmulHeuristicResult_t> candidates;
};

constexpr int kTuneCandidates = 24, kTuneRuns = 8;

std::mutex g_mu;
std::map<int, hipblasLtHandle_t> g_handles;                               // one handle per device (created on it)
hipblasLtHandle_t handle_of(int dev) {                                    // under g_mu
    auto it = g_handles.find(dev);
    if (it != g_handles.end()) return it->second;
    hipblasLtHandle_t h = nullptr;
    if (hipblasLtCreate(&h) != HIPBLAS_STATUS_SUCCESS) return nullptr;
    g_handles[dev] = h;
    return h;
}
std::map<std::tuple<int, int64_t, int, int, int, int>, GemmPlan> g_plans;   // (device, rows, Cin, Cout, relu, residual)

#define DMM_LT_TRY(expr)                                    \
    do {                                                    \
        if ((expr) != HIPBLAS_STATUS_SUCCESS) return false; \
    } while (0)

bool build_plan(hipblasLtHandle_t g_handle, GemmPlan &p, int64_t rows, int cin, int cout, bool relu, bool residual, size_t ws_limit) {
    const hipDataType bf16 = HIP_R_16BF;
    DMM_LT_TRY(hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    const hipblasOperation_t op_n = HIPBLAS_OP_N;
    DMM_LT_TRY(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &op_n, sizeof(op_n)));
    DMM_LT_TRY(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &op_n, sizeof(op_n)));
    const hipblasLtEpilogue_t epi = relu ? HIPBLASLT_EPILOGUE_RELU_BIAS : HIPBLASLT_EPILOGUE_BIAS;
    DMM_LT_TRY(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)));
    const int32_t bias_type = (int32_t)HIP_R_32F;
    DMM_LT_TRY(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bias_type, sizeof(bias_type)));
    // column-major problem: m = Cout, n = rows, k = Cin
    DMM_LT_TRY(hipblasLtMatrixLayoutCreate(&p.a, bf16, cout, cin, cout));        // W^T  [Cout, Cin], ld Cout
    DMM_LT_TRY(hipblasLtMatrixLayoutCreate(&p.b, bf16, cin, rows, cin));         // X^T  [Cin, rows], ld Cin
    DMM_LT_TRY(hipblasLtMatrixLayoutCreate(&p.c, bf16, cout, rows, cout));       // residual^T
    DMM_LT_TRY(hipblasLtMatrixLayoutCreate(&p.d, bf16, cout, rows, cout));       // Y^T
    hipblasLtMatmulPreference_t pref = nullptr;
    DMM_LT_TRY(hipblasLtMatmulPreferenceCreate(&pref));
    const uint64_t ws = ws_limit;
    hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws, sizeof(ws));
    // the heuristic needs the bias pointer attribute to be present for bias epilogues on some versions: a dummy
    // non-null value is enough for the query, the real pointer is set per call
    const void *dummy = (const void *)0x1000;
    hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &dummy, sizeof(dummy));
    const int want = [] {
        const int v = dmm::opt(DMM_OPT_GEMM_TUNE);
        return v < 1 ? 1 : (v > kTuneCandidates ? kTuneCandidates : v);
    }();
    hipblasLtMatmulHeuristicResult_t res[kTuneCandidates];
    int found = 0;
    const hipblasStatus_t st = hipblasLtMatmulAlgoGetHeuristic(g_handle, p.desc, p.a, p.b, residual ? p.c : p.d, p.d, pref,
                                                               want, res, &found);
    hipblasLtMatmulPreferenceDestroy(pref);
    if (st != HIPBLAS_STATUS_SUCCESS || found < 1) return false;
    p.algo = res[0].algo;
    p.workspace = res[0].workspaceSize;
    p.ok = true;
    for (int i = 0; i < found; ++i)
        if (res[i].state == HIPBLAS_STATUS_SUCCESS && res[i].workspaceSize <= ws_limit) p.candidates.push_back(res[i]);
    if (p.candidates.size() < 2) p.candidates.clear();
    return true;
}

// Time the heuristic's candidates on the operands of this call and keep the fastest (first un-captured call of a shape).
// Every candidate writes the same product into y, so the caller's result is unaffected; a candidate the library
// rejects at launch is skipped.  Caller holds g_mu.
void tune_plan(hipblasLtHandle_t g_handle, GemmPlan &p, const void *x, const void *w, const void *residual, void *y, void *workspace,
               size_t workspace_bytes, hipStream_t stream) {
    std::vector<hipblasLtMatmulHeuristicResult_t> cand;
    cand.swap(p.candidates);                                     // one attempt per plan, whatever happens
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess) return;
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return; }
    const float alpha = 1.0f, beta = residual ? 1.0f : 0.0f;
    float best = 0.0f;
    int best_i = -1;
    for (size_t i = 0; i < cand.size(); ++i) {
        bool ok = true;
        for (int r = 0; r < kTuneRuns + 1 && ok; ++r) {          // run 0 warms the kernel up (code load), then timed
            if (r == 1 && hipEventRecord(e0, stream) != hipSuccess) ok = false;
            ok = ok && hipblasLtMatmul(g_handle, p.desc, &alpha, w, p.a, x, p.b, &beta, residual ? residual : y,
                                       residual ? p.c : p.d, y, p.d, &cand[i].algo, workspace, workspace_bytes,
                                       stream) == HIPBLAS_STATUS_SUCCESS;
        }
        float ms = 0.0f;
        if (!ok || hipEventRecord(e1, stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
            hipEventElapsedTime(&ms, e0, e1) != hipSuccess) {
            (void)hipGetLastError();
            continue;
        }
        if (best_i < 0 || ms < best) { best = ms; best_i = (int)i; }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (best_i >= 0) {
        p.algo = cand[best_i].algo;
        p.workspace = cand[best_i].workspaceSize;
    }
}

}  // namespace
}  // namespace dmm

extern "C" int dmm_conv1x1_bf16(const void *x, const void *w, const float *bias, const void *residual, int64_t rows,
                                int cin, int cout, int relu, void *y, void *workspace, size_t workspace_bytes,
                                dmm_stream_t stream) {
    if (rows < 0 || cin <= 0 || cout <= 0) return DMM_ERR_BAD_ARG;
    if (rows == 0) return DMM_OK;
    if (!x || !w || !bias || !y) return DMM_ERR_BAD_ARG;
    int dev = 0;
    DMM_HIP_TRY(hipGetDevice(&dev));
    dmm::GemmPlan *plan = nullptr;
    hipblasLtHandle_t handle = nullptr;
    {
        std::lock_guard<std::mutex> lk(dmm::g_mu);
        handle = dmm::handle_of(dev);
        if (!handle) return DMM_ERR_LAUNCH;
        auto key = std::make_tuple(dev, rows, cin, cout, relu ? 1 : 0, residual ? 1 : 0);
        auto it = dmm::g_plans.find(key);
        if (it == dmm::g_plans.end()) {
            dmm::GemmPlan p;
            if (!dmm::build_plan(handle, p, rows, cin, cout, relu != 0, residual != nullptr, workspace ? workspace_bytes : 0))
                p.ok = false;
            it = dmm::g_plans.emplace(key, p).first;
        }
        plan = &it->second;
    }
    if (!plan->ok) return DMM_ERR_UNSUPPORTED;
    if (plan->workspace > (workspace ? workspace_bytes : 0)) return DMM_ERR_WORKSPACE;
    // the descriptor carries the bias pointer: set it under the lock-free assumption of one stream per plan user
    // (pointer attributes are plain fields read at launch time)
    std::lock_guard<std::mutex> lk(dmm::g_mu);
    const void *bp = bias;
    if (hipblasLtMatmulDescSetAttribute(plan->desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bp, sizeof(bp)) !=
        HIPBLAS_STATUS_SUCCESS)
        return DMM_ERR_LAUNCH;
    if (!plan->candidates.empty()) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone)
            dmm::tune_plan(handle, *plan, x, w, residual, y, workspace, workspace_bytes, (hipStream_t)stream);
        else
            (void)hipGetLastError();
    }
    const float alpha = 1.0f, beta = residual ? 1.0f : 0.0f;
    const hipblasStatus_t st =
        hipblasLtMatmul(handle, plan->desc, &alpha, w, plan->a, x, plan->b, &beta, residual ? residual : y,
                        residual ? plan->c : plan->d, y, plan->d, &plan->algo, workspace, workspace_bytes,
                        (hipStream_t)stream);
    if (st != HIPBLAS_STATUS_SUCCESS) {
        dmm::set_last_hip_error((int)st);
        return DMM_ERR_LAUNCH;
    }
    return DMM_OK;
}
