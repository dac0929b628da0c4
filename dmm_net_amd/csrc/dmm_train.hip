// dmm_train.hip -- the TRAINING call of the matching layer as two C-ABI entries.
//
// The reference's trainer calls MatchModel.forward once per (video, frame) with targets (dmm/modules/dmm_model.py:130-132)
// and autograd walks back through it.  Granular, that is ~9 library calls forward (dual counts, normalise x 2, cosine,
// solver, mix, IoU -> greedy one-hot -> mse as tensor ops) and 3 backward, each with its own output allocations on the
// host side: at one frame per call the HOST, not the device, set the pace of the drop-in.  Here:
//   dmm_match_train_forward   (5d)  cosine (+ table clear) -> counts against templates AND targets in one pass -> solver
//                                   -> mix -> matching loss, 5 launches (4 when a handful of dense frames take the front
//                                   kernel), everything the backward needs in caller buffers
//   dmm_match_train_backward  (5e)  normalise both feature sets (one launch) -> mix backward -> taped solver backward ->
//                                   feature-similarity backward, 4 launches
//   dmm_matching_loss_f32     (1e)  compute_matching_loss's tail on the device: gt IoU -> greedy one-hot -> mse
// Same kernels as the granular entries (bit-identical results); only the loss tail is new arithmetic here.
#include "dmm_common.h"
#include "dmm_solve.h"

namespace dmm {
int cosine_lanes_launch(const float *feat_t, const float *feat_p, int B, int N, int M, int D, float *cos_out,
                        hipStream_t stream, int32_t *zero_ptr, int64_t zero_words, const int32_t *n_valid);
int front_small_launch(const void *masks_p, const void *masks_t, const void *masks_t2, int dtype, const float *feat_t,
                       const float *feat_p, int B, int N, int M, int HW, int D, int64_t sp_b, int64_t sp_n, int64_t st_b,
                       int64_t st_m, int64_t st2_b, int64_t st2_m, float *cos_out, int32_t *inter, int32_t *area_p,
                       int32_t *area_t, int32_t *inter2, int32_t *area_t2, bool tables_zero, hipStream_t stream);
int iou_counts_prezeroed(const void *masks_p, const void *masks_t, int dtype, int B, int N, int M, int HW, int64_t sp_b,
                         int64_t sp_n, int64_t st_b, int64_t st_m, const int32_t *n_valid, const int32_t *m_valid,
                         int32_t *inter, int32_t *area_p, int32_t *area_t, dmm_stream_t stream);
int iou_counts_dual_prezeroed(const void *masks_p, const void *masks_t, const void *masks_t2, int dtype, int B, int N, int M,
                              int HW, int64_t sp_b, int64_t sp_n, int64_t st_b, int64_t st_m, int64_t st2_b, int64_t st2_m,
                              const int32_t *n_valid, const int32_t *m_valid, int32_t *inter, int32_t *area_p,
                              int32_t *area_t, int32_t *inter2, int32_t *area_t2, dmm_stream_t stream);
int feature_normalize2_launch(const float *in_a, int64_t rows_a, float *out_a, float *norms_a, const float *in_b,
                              int64_t rows_b, float *out_b, float *norms_b, int D, hipStream_t stream, void *zero_ptr = nullptr,
                              size_t zero_bytes = 0);
int mask_mix_bwd_prezeroed(const float *Rb, const void *masks_p, int dtype, const float *dout, int B, int N, int M, int Pp,
                           int HW, int64_t sp_b, int64_t sp_n, const int32_t *n_valid, const int32_t *m_valid, float *dRb,
                           dmm_stream_t stream);

// ---------------------------------------------------------------------------------------------------------------------
// compute_matching_loss after the counts (dmm/utils/match_helper.py:43-48):
//   gt_iou[m,n] = inter2 / (area_p[n] + area_t2[m] - inter2 + 1e-6)          compute_iou_binary_mask_2D, :9-28
//   gt = relax_matching(-gt_iou, 0, 0, 0)[0]  = the greedy one-hot init     relax_match.py:45-55
//        Cmax = max(C); per column the first argmin row keeps its value, the others become Cmax; per row the first
//        argmin column gets the 1
//   loss = mse(cos, gt) over the live block                                  F.mse_loss, :48
// One workgroup per frame, any M, N <= kLossMaxN (the per-column argmin rows sit in LDS).  The live block of a ragged
// frame is [m_valid, n_valid]; gt is written for all [M, N] slots (zeros outside the live block).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kLossThreads = 256;
constexpr int kLossMaxN = 8192, kLossMaxM = 4096;

__global__ __launch_bounds__(kLossThreads) void matching_loss_kernel(
    const int32_t *__restrict__ inter2, const int32_t *__restrict__ area_p, const int32_t *__restrict__ area_t2,
    const float *__restrict__ cosv, int N, int M, const int32_t *__restrict__ n_valid, const int32_t *__restrict__ m_valid,
    float *__restrict__ gt, float *__restrict__ loss) {
    extern __shared__ int loss_lds[];                        // colbest [N] | rowidx [M]
    __shared__ float red[kLossThreads / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nv = n_valid ? min(max(n_valid[b], 0), N) : N;
    const int mv = m_valid ? min(max(m_valid[b], 0), M) : M;
    const int32_t *gi = inter2 + (int64_t)b * M * N;
    const int32_t *ap = area_p + (int64_t)b * N;
    const int32_t *at = area_t2 + (int64_t)b * M;
    float *gt_b = gt + (int64_t)b * M * N;
    if (nv == 0 || mv == 0) {                                // dmm_model.py:118-122: a frame without templates has no loss
        for (int i = tid; i < M * N; i += kLossThreads) gt_b[i] = 0.0f;
        if (tid == 0) loss[b] = 0.0f;
        return;
    }
    int *colbest = loss_lds, *rowidx = loss_lds + N;
    auto C = [&](int i, int j) {
        const int32_t g = gi[(int64_t)i * N + j];
        const float u = (float)(ap[j] + at[i] - g) + 1e-6f;
        return -((float)g / u);
    };
    // per column: first argmin over the rows; and the maximum of the whole table
    float cmax = -__builtin_inff();
    for (int j = tid; j < nv; j += kLossThreads) {
        float best = C(0, j);
        int bi = 0;
        cmax = fmaxf(cmax, best);
        for (int i = 1; i < mv; ++i) {
            const float c = C(i, j);
            cmax = fmaxf(cmax, c);
            if (c < best) { best = c; bi = i; }
        }
        colbest[j] = bi;
    }
    cmax = wave_max(cmax);
    if (lane == 0) red[wave] = cmax;
    __syncthreads();
    cmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    // per row: first argmin over the columns of (C where the row won its column, Cmax elsewhere)
    for (int i = wave; i < mv; i += kLossThreads / 64) {
        float best = __builtin_inff();
        int bj = 0x7fffffff;
        for (int j = lane; j < nv; j += 64) {
            const float v = colbest[j] == i ? C(i, j) : cmax;
            if (v < best) { best = v; bj = j; }
        }
        const float mn = wave_min(best);
        const int idx = wave_min_i32(best == mn ? bj : 0x7fffffff);
        if (lane == 0) rowidx[i] = idx;
    }
    __syncthreads();
    float acc = 0.0f;
    const float *cos_b = cosv + (int64_t)b * M * N;
    for (int e = tid; e < M * N; e += kLossThreads) {
        const int i = e / N, j = e - i * N;
        const bool live = i < mv && j < nv;
        const float g = (live && rowidx[i] == j) ? 1.0f : 0.0f;
        gt_b[e] = g;
        if (live) {
            const float d = cos_b[e] - g;
            acc = __builtin_fmaf(d, d, acc);
        }
    }
    acc = wave_sum(acc);
    __syncthreads();                                         // red[] is read above by every wave
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid == 0) loss[b] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)((int64_t)nv * mv);
}

static inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }

struct TrainFwdWs {
    int32_t *inter, *area_p, *area_t, *inter2, *area_t2, *area_p2;
    float *featn_p, *featn_t;
    size_t table_words, bytes;
};
static TrainFwdWs carve_train_fwd(void *base, int B, int N, int M, int D) {
    TrainFwdWs w;
    size_t off = 0;
    auto take = [&](size_t n) {
        void *p = base ? (void *)((char *)base + off) : nullptr;
        off += align256(n);
        return p;
    };
    // inter | area_p | area_t | inter2 | area_t2 back to back: ONE clearing pass (the similarity launch does it on the side)
    w.table_words = 2 * ((size_t)B * M * N + (size_t)B * M) + (size_t)B * N;
    w.inter = (int32_t *)take(sizeof(int32_t) * w.table_words);
    w.area_p = w.inter ? w.inter + (size_t)B * M * N : nullptr;
    w.area_t = w.area_p ? w.area_p + (size_t)B * N : nullptr;
    w.inter2 = w.area_t ? w.area_t + (size_t)B * M : nullptr;
    w.area_t2 = w.inter2 ? w.inter2 + (size_t)B * M * N : nullptr;
    // > 16 rows per set: the targets take a count pass of their own, whose proposal areas (the same numbers) land here
    w.area_p2 = (int32_t *)take(M > 16 ? sizeof(int32_t) * (size_t)B * N : 0);
    // the normalised rows are only needed when the similarity does not take the one-launch kernel (other D, N = 1)
    w.featn_p = (float *)take(sizeof(float) * (size_t)B * N * D);
    w.featn_t = (float *)take(sizeof(float) * (size_t)B * M * D);
    w.bytes = off;
    return w;
}

struct TrainBwdWs {
    float *featn_p, *featn_t, *norm_p, *norm_t, *dRb, *dsim;
    void *tape;
    size_t tape_bytes, bytes;
};
static TrainBwdWs carve_train_bwd(void *base, int B, int N, int M, int D, int max_iter, int proj_iter) {
    const int Pp = N > M ? N : M + 1;
    TrainBwdWs w;
    size_t off = 0;
    auto take = [&](size_t n) {
        void *p = base ? (void *)((char *)base + off) : nullptr;
        off += align256(n);
        return p;
    };
    w.featn_p = (float *)take(sizeof(float) * (size_t)B * N * D);
    w.featn_t = (float *)take(sizeof(float) * (size_t)B * M * D);
    w.norm_p = (float *)take(sizeof(float) * (size_t)B * N);
    w.norm_t = (float *)take(sizeof(float) * (size_t)B * M);
    w.dRb = (float *)take(sizeof(float) * (size_t)B * M * Pp);
    w.dsim = (float *)take(sizeof(float) * (size_t)B * M * N);
    w.tape_bytes = dmm_relax_bwd_workspace_bytes(B, N, M, max_iter, proj_iter);
    w.tape = take(w.tape_bytes);
    w.bytes = off;
    return w;
}
}  // namespace dmm

// (1e)
extern "C" int dmm_matching_loss_f32(const int32_t *inter2, const int32_t *area_p, const int32_t *area_t2, const float *cosv,
                                     int B, int N, int M, const int32_t *n_valid, const int32_t *m_valid, float *gt,
                                     float *loss, dmm_stream_t stream) {
    if (B < 0 || N < 0 || M < 0) return DMM_ERR_BAD_ARG;
    if (B == 0) return DMM_OK;
    if (!loss) return DMM_ERR_BAD_ARG;
    if (N == 0 || M == 0) {                                  // no table: zero loss, nothing to write into gt
        DMM_HIP_TRY(dmm::zero_async(loss, sizeof(float) * (size_t)B, (hipStream_t)stream));
        return DMM_OK;
    }
    if (!inter2 || !area_p || !area_t2 || !cosv || !gt) return DMM_ERR_BAD_ARG;
    if (N > dmm::kLossMaxN || M > dmm::kLossMaxM || (int64_t)M * N > 0x7fffffffLL) return DMM_ERR_UNSUPPORTED;
    const size_t lds = sizeof(int) * ((size_t)N + M);
    hipLaunchKernelGGL(dmm::matching_loss_kernel, dim3(B), dim3(dmm::kLossThreads), lds, (hipStream_t)stream, inter2, area_p,
                       area_t2, cosv, N, M, n_valid, m_valid, gt, loss);
    return dmm::check_launch();
}

// The solver tape of the training call: the forward's one-wave solver kernel records, per projection sweep and column, which
// rows passed the relu and whether the column sum exceeded 1 (8 bytes), per outer iteration how many sweeps ran, and R; given
// those a sweep is a linear map, so the backward walks the records instead of re-running the solver to rebuild them (one
// 5 x 50 frame at 10 x 5: 48 -> ~24 us of a 186 us forward + backward call).  Block = R [B, M, Pp] | records | sweep counts.
namespace dmm {
static size_t train_tape_r_bytes(int B, int N, int M) {
    return align256(sizeof(float) * (size_t)B * M * (size_t)(N > M ? N : M + 1));
}
}  // namespace dmm
extern "C" size_t dmm_match_train_tape_bytes(int B, int N, int M, int max_iter, int proj_iter) {
    const size_t t = dmm::relax_tape_bytes(B, N, M, max_iter, proj_iter);     // 0: tables the one-wave kernels do not take
    return t ? dmm::train_tape_r_bytes(B, N, M) + t : 0;
}

// (5d)
extern "C" size_t dmm_match_train_forward_workspace_bytes(int B, int N, int M, int D) {
    if (B <= 0 || N <= 0 || M <= 0 || D < 0) return 0;
    return dmm::carve_train_fwd(nullptr, B, N, M, D).bytes;
}

extern "C" int dmm_match_train_forward(const void *masks_p, const void *masks_t, const void *targets, int mask_dtype,
                                       const float *feat_p, const float *feat_t, const float *score_p, int B, int N, int M,
                                       int HW, int D, int64_t sp_b, int64_t sp_n, int64_t st_b, int64_t st_m, int64_t sg_b,
                                       int64_t sg_m, const int32_t *n_valid, const int32_t *m_valid, float score_weight,
                                       int max_iter, int proj_iter, float lr, int is_test, float *full_outmask,
                                       float *match_score, float *det_score, float *cost_loss, int32_t *iters_out,
                                       float *cos_out, float *sim_out, float *Rb_out, float *gt_out, void *workspace,
                                       size_t workspace_bytes, void *tape, size_t tape_bytes, int *taped,
                                       dmm_stream_t stream) {
    if (taped) *taped = 0;
    if (B < 0 || N < 0 || M < 0 || HW < 0 || D < 0 || max_iter < 0 || proj_iter < 0) return DMM_ERR_BAD_ARG;
    if (B == 0 || M == 0) return DMM_OK;
    if (N == 0) return DMM_ERR_BAD_ARG;
    if (!masks_p || !masks_t || !feat_p || !feat_t || !score_p || !full_outmask || !match_score || !det_score || !cos_out ||
        !sim_out || !Rb_out || !workspace)
        return DMM_ERR_BAD_ARG;
    if (targets && (!cost_loss || !gt_out)) return DMM_ERR_BAD_ARG;
    if (mask_dtype != DMM_F32 && mask_dtype != DMM_F16 && mask_dtype != DMM_BF16) return DMM_ERR_BAD_ARG;
    const int Pp = N > M ? N : M + 1;
    // the fast kernels' envelope; wider tables train through the granular entries (any size)
    if (M > DMM_MAX_TEMPLATES || Pp > DMM_MAX_PROPOSALS || B > 65535 || dmm::opt(DMM_OPT_FORCE_WIDE) == 1)
        return DMM_ERR_UNSUPPORTED;
    dmm::TrainFwdWs w = dmm::carve_train_fwd(workspace, B, N, M, D);
    if (workspace_bytes < w.bytes) return DMM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const bool dense = !n_valid && !m_valid;
    // one pass over the proposal planes for both IoU tables while both template sets fit one tile (<= 16 rows each)
    const bool dual = targets && M <= 16;
    const bool force_tile = dmm::opt(DMM_OPT_COSINE_KERNEL) == 1;
    int rc = DMM_ERR_UNSUPPORTED;
    bool counted = false;
    if (dense && !force_tile && (dual || !targets)) {      // (proposal planes behind a pointer table included: frame_base)
        // a handful of dense frames: similarity and counts beside each other in ONE launch (behind one clearing launch)
        rc = dmm::front_small_launch(masks_p, masks_t, dual ? targets : nullptr, mask_dtype, feat_t, feat_p, B, N, M, HW, D,
                                     sp_b, sp_n, st_b, st_m, sg_b, sg_m, cos_out, w.inter, w.area_p, w.area_t,
                                     dual ? w.inter2 : nullptr, dual ? w.area_t2 : nullptr, false, s);
        if (rc == DMM_OK) counted = true;
        else if (rc != DMM_ERR_UNSUPPORTED) return rc;
    }
    if (!counted) {
        // the one-launch similarity kernel, which also clears the five count tables: every frame in the summation order of
        // ITS live proposal count; template rows past m_valid are computed and never read (every consumer masks them)
        rc = force_tile ? DMM_ERR_UNSUPPORTED
                        : dmm::cosine_lanes_launch(feat_t, feat_p, B, N, M, D, cos_out, s, w.inter, (int64_t)w.table_words,
                                                   n_valid);
        if (rc != DMM_OK && rc != DMM_ERR_UNSUPPORTED) return rc;
        const bool zeroed = rc == DMM_OK;
        if (!zeroed) {
            DMM_HIP_TRY(dmm::zero_async(w.inter, sizeof(int32_t) * w.table_words, s));
            rc = dense ? dmm_cosine_features_f32(feat_t, feat_p, B, N, M, D, cos_out, stream) : DMM_ERR_UNSUPPORTED;
            if (rc == DMM_ERR_UNSUPPORTED) {
                rc = dmm::feature_normalize2_launch(feat_p, (int64_t)B * N, w.featn_p, nullptr, feat_t, (int64_t)B * M,
                                                    w.featn_t, nullptr, D, s);
                if (rc != DMM_OK) return rc;
                rc = dmm_cosine_f32(w.featn_t, w.featn_p, B, N, M, D, n_valid, m_valid, cos_out, stream);
            }
            if (rc != DMM_OK) return rc;
        }
        if (dual) {
            rc = dmm::iou_counts_dual_prezeroed(masks_p, masks_t, targets, mask_dtype, B, N, M, HW, sp_b, sp_n, st_b, st_m,
                                                sg_b, sg_m, n_valid, m_valid, w.inter, w.area_p, w.area_t, w.inter2,
                                                w.area_t2, stream);
        } else {
            rc = dmm::iou_counts_prezeroed(masks_p, masks_t, mask_dtype, B, N, M, HW, sp_b, sp_n, st_b, st_m, n_valid, m_valid,
                                           w.inter, w.area_p, w.area_t, stream);
        }
        if (rc != DMM_OK) return rc;
    }
    if (targets && !dual) {
        // > 16 rows per set: the targets in a count pass of their own
        rc = dmm_iou_counts(masks_p, targets, mask_dtype, B, N, M, HW, sp_b, sp_n, sg_b, sg_m, n_valid, m_valid, w.inter2,
                            w.area_p2, w.area_t2, stream);
        if (rc != DMM_OK) return rc;
    }
    {
        // keep the solver's tape for the backward when the caller gave room for it (and asked for the iteration counts)
        const size_t need = dmm_match_train_tape_bytes(B, N, M, max_iter, proj_iter);
        const bool keep = tape && taped && iters_out && need > 0 && tape_bytes >= need;
        float *R_keep = keep ? (float *)tape : nullptr;
        void *records = keep ? (void *)((char *)tape + dmm::train_tape_r_bytes(B, N, M)) : nullptr;
        rc = dmm::relax_match_launch(cos_out, w.inter, w.area_p, w.area_t, score_p, B, N, M, n_valid, m_valid, score_weight,
                                     max_iter, proj_iter, lr, is_test, sim_out, R_keep, Rb_out, match_score, det_score,
                                     iters_out, nullptr, 0, nullptr, stream, records, keep ? taped : nullptr);
    }
    if (rc != DMM_OK) return rc;
    // train mode keeps every R > 0.01: the rows share planes -> the union of the supports is streamed once
    if (!is_test)
        rc = dmm_mask_mix_shared_to(Rb_out, masks_p, mask_dtype, B, N, M, Pp, HW, sp_b, sp_n, n_valid, m_valid, full_outmask,
                                    DMM_F32, (int64_t)M * HW, HW, stream);
    else
        rc = dmm_mask_mix(Rb_out, masks_p, mask_dtype, B, N, M, Pp, HW, sp_b, sp_n, n_valid, m_valid, full_outmask,
                          (int64_t)M * HW, HW, stream);
    if (rc != DMM_OK || !targets) return rc;
    return dmm_matching_loss_f32(w.inter2, w.area_p, w.area_t2, cos_out, B, N, M, n_valid, m_valid, gt_out, cost_loss,
                                 stream);
}

// (5e)
extern "C" size_t dmm_match_train_backward_workspace_bytes(int B, int N, int M, int D, int max_iter, int proj_iter) {
    if (B <= 0 || N <= 0 || M <= 0 || D <= 0 || max_iter < 0 || proj_iter < 0) return 0;
    return dmm::carve_train_bwd(nullptr, B, N, M, D, max_iter, proj_iter).bytes;
}

extern "C" int dmm_match_train_backward(const void *masks_p, int mask_dtype, const float *feat_p, const float *feat_t,
                                        const float *score_p, const float *cosv, const float *sim, const float *Rb,
                                        const float *gt, const float *d_full, const float *d_match_score,
                                        const float *d_det_score, const float *d_loss, int B, int N, int M, int HW, int D,
                                        int64_t sp_b, int64_t sp_n, const int32_t *n_valid, const int32_t *m_valid,
                                        float score_weight, int max_iter, int proj_iter, float lr, int is_test,
                                        float *g_feat_t, float *g_feat_p, void *workspace, size_t workspace_bytes,
                                        const void *tape, const int32_t *iters, int taped, dmm_stream_t stream) {
    if (B < 0 || N < 0 || M < 0 || HW < 0 || D < 0 || max_iter < 0 || proj_iter < 0) return DMM_ERR_BAD_ARG;
    if (B == 0 || M == 0 || D == 0) return DMM_OK;
    if (N == 0) return DMM_ERR_BAD_ARG;
    if (!feat_p || !feat_t || !score_p || !sim || !Rb || !g_feat_t || !g_feat_p || !workspace) return DMM_ERR_BAD_ARG;
    if (d_full && !masks_p) return DMM_ERR_BAD_ARG;
    if ((gt || d_loss) && !(gt && d_loss && cosv)) return DMM_ERR_BAD_ARG;
    const int Pp = N > M ? N : M + 1;
    dmm::TrainBwdWs w = dmm::carve_train_bwd(workspace, B, N, M, D, max_iter, proj_iter);
    if (workspace_bytes < w.bytes) return DMM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    // (the normalising launch also clears dRb for the mix backward behind it: one launch less in a one-frame call)
    int rc = dmm::feature_normalize2_launch(feat_p, (int64_t)B * N, w.featn_p, w.norm_p, feat_t, (int64_t)B * M, w.featn_t,
                                            w.norm_t, D, s, d_full ? (void *)w.dRb : nullptr,
                                            sizeof(float) * (size_t)B * M * Pp);
    if (rc != DMM_OK) return rc;
    const float *dRb = nullptr;
    if (d_full) {
        rc = dmm::mask_mix_bwd_prezeroed(Rb, masks_p, mask_dtype, d_full, B, N, M, Pp, HW, sp_b, sp_n, n_valid, m_valid, w.dRb,
                                         stream);
        if (rc != DMM_OK) return rc;
        dRb = w.dRb;
    }
    {
        // taped = what dmm_match_train_forward returned through *taped for THIS tape block: walk the forward's records
        const bool walk = taped && tape && iters && dmm_match_train_tape_bytes(B, N, M, max_iter, proj_iter) > 0;
        const void *records = walk ? (const void *)((const char *)tape + dmm::train_tape_r_bytes(B, N, M)) : nullptr;
        rc = dmm::relax_match_bwd_launch(sim, score_p, B, N, M, n_valid, m_valid, max_iter, proj_iter, lr, is_test, dRb,
                                         d_match_score, d_det_score, w.dsim, w.tape, w.tape_bytes, records,
                                         walk ? (const float *)tape : nullptr, walk ? iters : nullptr, stream);
    }
    if (rc != DMM_OK) return rc;
    return dmm_feature_sim_bwd_f32(w.dsim, cosv, gt, d_loss, score_weight, feat_t, feat_p, w.featn_t, w.featn_p, w.norm_t,
                                   w.norm_p, B, N, M, D, n_valid, m_valid, g_feat_t, g_feat_p, stream);
}
