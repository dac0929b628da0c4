// dmm_api.hip -- C-ABI glue of libdmm_match.so: status/reporting and the fused forward entry point
// that chains the four kernels of MatchModel.forward (dmm/modules/match_model.py:24-47) on one stream.
#include <limits.h>
#include <stdlib.h>

#include <atomic>

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
static thread_local int g_last_hip_error = 0;
void set_last_hip_error(int e) { g_last_hip_error = e; }
static std::atomic<long long> g_launches{0};
void note_launch() { g_launches.fetch_add(1, std::memory_order_relaxed); }

// ---- dispatch options -------------------------------------------------------------------------------------------------
static constexpr int kOptDefaults[DMM_OPT_COUNT] = {
    /* COST_KERNEL */ -1,      /* COST_TINY_FRAMES */ 8,   /* SOLVER_KERNEL */ -1,  /* FORCE_WIDE */ 0,
    /* COSINE_KERNEL */ 0,     /* COST_WGS */ 8192,        /* COST_SMALL_WGS */ 512, /* COST_TL_WGS */ 512,
    /* COST_XCD */ 1,          /* MIX_XCD */ 3,            /* MIX_WGS */ 320000,    /* MIX_STEPQ */ 2,
    /* MIX_ALIGN */ 128,       /* MIX_NT */ 3,             /* SOLVER_HELPER_MAX */ 512, /* NMS_WAVE */ 1,
    /* COS_ROWS_MIN_N */ 65,   /* GEMM_TUNE */ 1,          /* PACK_VARIANT */ 4,    /* SMALL_FUSED */ 1,
    /* MIX_SHARED */ -1,       /* MIX_SHARED_STEPS */ 1,   /* FEAT_BWD_FRAME */ -1, /* MIX_SHARED_LOCKSTEP */ 1,
};
static std::atomic<int> g_opts[DMM_OPT_COUNT] = {
    {kOptDefaults[0]},  {kOptDefaults[1]},  {kOptDefaults[2]},  {kOptDefaults[3]},  {kOptDefaults[4]},
    {kOptDefaults[5]},  {kOptDefaults[6]},  {kOptDefaults[7]},  {kOptDefaults[8]},  {kOptDefaults[9]},
    {kOptDefaults[10]}, {kOptDefaults[11]}, {kOptDefaults[12]}, {kOptDefaults[13]}, {kOptDefaults[14]},
    {kOptDefaults[15]}, {kOptDefaults[16]}, {kOptDefaults[17]}, {kOptDefaults[18]}, {kOptDefaults[19]},
    {kOptDefaults[20]}, {kOptDefaults[21]}, {kOptDefaults[22]}, {kOptDefaults[23]},
};
static_assert(DMM_OPT_COUNT == 24, "kOptDefaults / g_opts list every option");
int opt(int key) { return g_opts[key].load(std::memory_order_relaxed); }

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Workspace {
    int32_t *inter, *area_p, *area_t;
    float *featn_p, *featn_t, *cosv, *sim, *Rb;
    float *wide;                       // solver state of the general kernel (tables outside the compiled envelope)
    size_t bytes;
};

// tables the fast solver kernels are not compiled for: the general forms of dmm_wide.hip take them
// (DMM_OPT_FORCE_WIDE: tests send tables inside the envelope there too)
static bool wide_shape(int N, int M) {
    const int Pp = N > M ? N : M + 1;
    return M > DMM_MAX_TEMPLATES || Pp > DMM_MAX_PROPOSALS || opt(DMM_OPT_FORCE_WIDE) == 1;
}

static Workspace carve(void *base, int B, int N, int M, int D) {
    const int Pp = N > M ? N : M + 1;
    Workspace w;
    size_t off = 0;
    auto take = [&](size_t n) {
        void *p = base ? (void *)((char *)base + off) : nullptr;
        off += align_up(n, 256);
        return p;
    };
    // inter | area_p | area_t back to back: dmm_iou_counts clears them with one memset
    w.inter = (int32_t *)take(sizeof(int32_t) * ((size_t)B * M * N + (size_t)B * N + (size_t)B * M));
    w.area_p = w.inter ? w.inter + (size_t)B * M * N : nullptr;
    w.area_t = w.area_p ? w.area_p + (size_t)B * N : nullptr;
    w.featn_p = (float *)take(sizeof(float) * (size_t)B * N * D);
    w.featn_t = (float *)take(sizeof(float) * (size_t)B * M * D);
    w.cosv = (float *)take(sizeof(float) * (size_t)B * M * N);
    w.sim = (float *)take(sizeof(float) * (size_t)B * M * N);
    w.Rb = (float *)take(sizeof(float) * (size_t)B * M * Pp);
    w.wide = nullptr;
    if (wide_shape(N, M))                                        // the one test both the sizes and the dispatch use
        w.wide = (float *)take(sizeof(float) * (size_t)B * wide_scratch_floats(M, Pp));
    w.bytes = off;
    return w;
}
}  // namespace dmm

extern "C" int dmm_abi_version(void) { return DMM_ABI_VERSION; }

extern "C" int dmm_set_option(int option, int value) {
    if (option < 0 || option >= DMM_OPT_COUNT) return DMM_ERR_BAD_ARG;
    switch (option) {                                            // ranges: a bad value must not reach a launch computation
        case DMM_OPT_FEAT_BWD_FRAME: if (value < -1 || value > 2) return DMM_ERR_BAD_ARG; break;
        case DMM_OPT_COST_KERNEL: case DMM_OPT_SOLVER_KERNEL: case DMM_OPT_MIX_SHARED:
            if (value < -1 || value > 1) return DMM_ERR_BAD_ARG; break;
        case DMM_OPT_MIX_XCD: if (value < 0 || value > 7) return DMM_ERR_BAD_ARG; break;
        case DMM_OPT_FORCE_WIDE: case DMM_OPT_COSINE_KERNEL: case DMM_OPT_COST_XCD:
        case DMM_OPT_NMS_WAVE: case DMM_OPT_SMALL_FUSED: case DMM_OPT_MIX_SHARED_LOCKSTEP:
            if (value < 0 || value > 1) return DMM_ERR_BAD_ARG; break;
        case DMM_OPT_MIX_ALIGN: if (value != 16 && value != 32 && value != 64 && value != 128) return DMM_ERR_BAD_ARG; break;
        case DMM_OPT_MIX_NT: if (value < 0 || value > 3) return DMM_ERR_BAD_ARG; break;
        case DMM_OPT_COST_WGS: case DMM_OPT_COST_SMALL_WGS: case DMM_OPT_COST_TL_WGS: case DMM_OPT_MIX_WGS:
        case DMM_OPT_MIX_STEPQ: case DMM_OPT_GEMM_TUNE: case DMM_OPT_COS_ROWS_MIN_N: case DMM_OPT_MIX_SHARED_STEPS:
            if (value < 1) return DMM_ERR_BAD_ARG; break;
        default: if (value < 0) return DMM_ERR_BAD_ARG; break;
    }
    dmm::g_opts[option].store(value, std::memory_order_relaxed);
    return DMM_OK;
}

extern "C" int dmm_get_option(int option) {
    return (option < 0 || option >= DMM_OPT_COUNT) ? INT_MIN : dmm::opt(option);
}

extern "C" int dmm_reset_options(void) {
    for (int k = 0; k < DMM_OPT_COUNT; ++k) dmm::g_opts[k].store(dmm::kOptDefaults[k], std::memory_order_relaxed);
    return DMM_OK;
}

extern "C" const char *dmm_status_string(int status) {
    switch (status) {
        case DMM_OK: return "ok";
        case DMM_ERR_BAD_ARG: return "bad argument";
        case DMM_ERR_UNSUPPORTED: return "shape outside the compiled kernel envelope";
        case DMM_ERR_LAUNCH: return "HIP launch/runtime error (see dmm_last_hip_error)";
        case DMM_ERR_WORKSPACE: return "workspace too small";
        default: return "unknown status";
    }
}

extern "C" int dmm_last_hip_error(void) { return dmm::g_last_hip_error; }

extern "C" long long dmm_launch_count(void) { return dmm::g_launches.load(std::memory_order_relaxed); }

#define DMM_STR_(x) #x
#define DMM_STR(x) DMM_STR_(x)
extern "C" const char *dmm_build_info(void) { return "libdmm_match gfx950 (CDNA4, wave64) abi " DMM_STR(DMM_ABI_VERSION); }

extern "C" size_t dmm_workspace_bytes(int B, int N, int M, int D) {
    if (B <= 0 || N <= 0 || M <= 0 || D < 0) return 0;
    return dmm::carve(nullptr, B, N, M, D).bytes;
}

extern "C" int dmm_match_forward(const void *masks_p, const void *masks_t, int mask_dtype, const float *feat_p,
                                 const float *feat_t, const float *score_p, int B, int N, int M, int HW, int D,
                                 int64_t sp_b, int64_t sp_n, int64_t st_b, int64_t st_m, const int32_t *n_valid,
                                 const int32_t *m_valid, float score_weight, int max_iter, int proj_iter, float lr,
                                 int is_test, float *full_outmask, float *match_score, float *det_score,
                                 float *sim_out, float *R_out, float *Rb_out, int32_t *iters_out, void *workspace,
                                 size_t workspace_bytes, dmm_stream_t stream) {
    return dmm_match_forward_ws(masks_p, masks_t, mask_dtype, feat_p, feat_t, score_p, B, N, M, HW, D, sp_b, sp_n, st_b, st_m,
                                n_valid, m_valid, score_weight, max_iter, proj_iter, lr, is_test, full_outmask, match_score,
                                det_score, sim_out, R_out, Rb_out, iters_out, workspace, workspace_bytes, nullptr, stream);
}

// (5a') dmm_match_forward for a caller that keeps ONE workspace for a sequence of calls and lets the library remember
// what it left there.  *ws_state in: DMM_WS_TABLES_ZERO if the previous call on this workspace (same B, N, M) returned
// that value, DMM_WS_UNKNOWN otherwise; out: the state the work enqueued by this call leaves behind.  A handful of dense
// frames (the small-batch front kernel) then run without the clearing launch in front of the counts -- a launch is
// ~4.5 us of a ~95 us one-frame call: the solver zeroes every table entry right after reading it.  Every other path
// returns DMM_WS_UNKNOWN.  On an error return the state is DMM_WS_UNKNOWN.
extern "C" int dmm_match_forward_ws(const void *masks_p, const void *masks_t, int mask_dtype, const float *feat_p,
                                    const float *feat_t, const float *score_p, int B, int N, int M, int HW, int D,
                                    int64_t sp_b, int64_t sp_n, int64_t st_b, int64_t st_m, const int32_t *n_valid,
                                    const int32_t *m_valid, float score_weight, int max_iter, int proj_iter, float lr,
                                    int is_test, float *full_outmask, float *match_score, float *det_score,
                                    float *sim_out, float *R_out, float *Rb_out, int32_t *iters_out, void *workspace,
                                    size_t workspace_bytes, int *ws_state, dmm_stream_t stream) {
    const bool tables_zero = ws_state && *ws_state == DMM_WS_TABLES_ZERO;
    if (ws_state) *ws_state = DMM_WS_UNKNOWN;
    int cleared = 0;
    if (B < 0 || N < 0 || M < 0 || HW < 0 || D < 0) return DMM_ERR_BAD_ARG;
    if (B == 0 || M == 0) return DMM_OK;
    if (N == 0) return DMM_ERR_BAD_ARG;
    if (!masks_p || !masks_t || !feat_p || !feat_t || !score_p || !full_outmask || !match_score || !det_score ||
        !workspace)
        return DMM_ERR_BAD_ARG;
    const int Pp = N > M ? N : M + 1;
    // the mix needs the pixel values: 1-bit planes are an input format of dmm_iou_counts only (reject before any launch)
    if (mask_dtype != DMM_F32 && mask_dtype != DMM_F16 && mask_dtype != DMM_BF16) return DMM_ERR_BAD_ARG;
    dmm::Workspace w = dmm::carve(workspace, B, N, M, D);
    if (workspace_bytes < w.bytes) return DMM_ERR_WORKSPACE;
    float *sim = sim_out ? sim_out : w.sim;
    float *Rb = Rb_out ? Rb_out : w.Rb;
    if (dmm::wide_shape(N, M)) {
        // Outside the envelope of the fast kernels (M <= 32, Pp <= 256): counts (they tile any N x M), the features
        // normalised, then the general kernels -- same operations, same order, any size (dmm_wide.hip).
        if (!w.wide || B > 65535) return DMM_ERR_UNSUPPORTED;
        int rc = dmm_iou_counts(masks_p, masks_t, mask_dtype, B, N, M, HW, sp_b, sp_n, st_b, st_m, n_valid, m_valid, w.inter,
                                w.area_p, w.area_t, stream);
        if (rc != DMM_OK) return rc;
        rc = dmm_feature_normalize_f32(feat_p, (int64_t)B * N, D, w.featn_p, nullptr, stream);
        if (rc != DMM_OK) return rc;
        rc = dmm_feature_normalize_f32(feat_t, (int64_t)B * M, D, w.featn_t, nullptr, stream);
        if (rc != DMM_OK) return rc;
        rc = dmm::launch_cosine_wide(w.featn_t, w.featn_p, B, N, M, D, n_valid, m_valid, w.cosv, (hipStream_t)stream);
        if (rc != DMM_OK) return rc;
        if (max_iter < 0 || proj_iter < 0) return DMM_ERR_BAD_ARG;
        const float w_feat = (float)(1.0 - (double)score_weight);
        rc = dmm::launch_relax_match_wide(w.cosv, w.inter, w.area_p, w.area_t, score_p, B, N, M, n_valid, m_valid, w_feat,
                                          score_weight, dmm::RelaxParams{max_iter, proj_iter, lr}, is_test, sim, R_out, Rb,
                                          match_score, det_score, iters_out, nullptr, w.wide, (hipStream_t)stream);
        if (rc != DMM_OK) return rc;
        return dmm_mask_mix(Rb, masks_p, mask_dtype, B, N, M, Pp, HW, sp_b, sp_n, n_valid, m_valid, full_outmask,
                            (int64_t)M * HW, HW, stream);
    }
    // Feature similarity FIRST when the batch is dense and D is one the lanes kernel takes: that launch also clears the
    // three count tables (contiguous in the workspace), so the counts start without a memset node -- 4.6 us of a
    // one-frame call's 135.  Otherwise counts (with their memset), then the tile kernel or normalise x 2 + cosine.
    const bool force_tile = dmm::opt(DMM_OPT_COSINE_KERNEL) == 1;
    int rc = DMM_ERR_UNSUPPORTED;
    // a handful of dense frames: table clear, then similarity and counts beside each other in ONE launch
    if (!n_valid && !m_valid && !force_tile) {
        rc = dmm::front_small_launch(masks_p, masks_t, nullptr, mask_dtype, feat_t, feat_p, B, N, M, HW, D, sp_b, sp_n, st_b,
                                     st_m, 0, 0, w.cosv, w.inter, w.area_p, w.area_t, nullptr, nullptr, tables_zero,
                                     (hipStream_t)stream);
        if (rc == DMM_OK) {
            // the solver reads the tables and (asked to) leaves them zero for the next call on this workspace
            rc = dmm::relax_match_launch(w.cosv, w.inter, w.area_p, w.area_t, score_p, B, N, M, n_valid, m_valid,
                                         score_weight, max_iter, proj_iter, lr, is_test, sim, R_out, Rb, match_score,
                                         det_score, iters_out, nullptr, ws_state != nullptr, &cleared, stream);
            if (rc != DMM_OK) return rc;
            goto mix;
        }
        if (rc != DMM_ERR_UNSUPPORTED) return rc;
    }
    if (!n_valid && !m_valid && !force_tile)
        rc = dmm::cosine_lanes_launch(feat_t, feat_p, B, N, M, D, w.cosv, (hipStream_t)stream, w.inter,
                                      (int64_t)B * M * N + (int64_t)B * N + (int64_t)B * M, nullptr);
    if (rc == DMM_OK) {
        rc = dmm::iou_counts_prezeroed(masks_p, masks_t, mask_dtype, B, N, M, HW, sp_b, sp_n, st_b, st_m, n_valid, m_valid,
                                       w.inter, w.area_p, w.area_t, stream);
        if (rc != DMM_OK) return rc;
    } else if (rc != DMM_ERR_UNSUPPORTED) {
        return rc;
    } else {
        rc = dmm_iou_counts(masks_p, masks_t, mask_dtype, B, N, M, HW, sp_b, sp_n, st_b, st_m, n_valid, m_valid, w.inter,
                            w.area_p, w.area_t, stream);
        if (rc != DMM_OK) return rc;
        rc = (!n_valid && !m_valid) ? dmm_cosine_features_f32(feat_t, feat_p, B, N, M, D, w.cosv, stream)
                                    : DMM_ERR_UNSUPPORTED;
    }
    if (rc == DMM_ERR_UNSUPPORTED) {
        rc = dmm_feature_normalize_f32(feat_p, (int64_t)B * N, D, w.featn_p, nullptr, stream);
        if (rc != DMM_OK) return rc;
        rc = dmm_feature_normalize_f32(feat_t, (int64_t)B * M, D, w.featn_t, nullptr, stream);
        if (rc != DMM_OK) return rc;
        rc = dmm_cosine_f32(w.featn_t, w.featn_p, B, N, M, D, n_valid, m_valid, w.cosv, stream);
    }
    if (rc != DMM_OK) return rc;
    rc = dmm_relax_match_f32(w.cosv, w.inter, w.area_p, w.area_t, score_p, B, N, M, n_valid, m_valid, score_weight,
                             max_iter, proj_iter, lr, is_test, sim, R_out, Rb, match_score, det_score, iters_out,
                             nullptr, stream);
    if (rc != DMM_OK) return rc;
mix:
    // train mode keeps every R > 0.01: the rows share planes -> the union of the supports is streamed once
    if (!is_test)
        rc = dmm_mask_mix_shared_to(Rb, masks_p, mask_dtype, B, N, M, Pp, HW, sp_b, sp_n, n_valid, m_valid, full_outmask,
                                    DMM_F32, (int64_t)M * HW, HW, stream);
    else
        rc = dmm_mask_mix(Rb, masks_p, mask_dtype, B, N, M, Pp, HW, sp_b, sp_n, n_valid, m_valid, full_outmask,
                          (int64_t)M * HW, HW, stream);
    if (rc == DMM_OK && ws_state && cleared) *ws_state = DMM_WS_TABLES_ZERO;
    return rc;
}

// ---------------------------------------------------------------------------------------------
// (5b) The same forward with the proposal side of the cost pass on 1-bit planes (DMM_PACKED1) that the caller already
// holds -- dmm_paste_masks_f32 / dmm_paste_kept_f32 emit them next to the soft planes.  The M template planes of each
// frame are packed here (one read of them, what the float count kernel would have read anyway), the counts run on the
// words (1/32 of the proposal bytes, identical integer tables), the mix reads the soft planes.
// ---------------------------------------------------------------------------------------------
static size_t packed_t_bytes(int B, int M, int HW) {
    return dmm::align_up(sizeof(uint64_t) * (size_t)B * M * (size_t)dmm_pack_words(HW), 256);
}

extern "C" size_t dmm_workspace_bytes_packed(int B, int N, int M, int D, int HW) {
    if (B <= 0 || N <= 0 || M <= 0 || D < 0 || HW < 0) return 0;
    return dmm::carve(nullptr, B, N, M, D).bytes + packed_t_bytes(B, M, HW);
}

extern "C" int dmm_match_forward_packed(const void *masks_p, const uint64_t *packed_p, const void *masks_t, int mask_dtype,
                                        const float *feat_p, const float *feat_t, const float *score_p, int B, int N, int M,
                                        int HW, int D, int64_t sp_b, int64_t sp_n, int64_t pk_b, int64_t pk_n, int64_t st_b,
                                        int64_t st_m, const int32_t *n_valid, const int32_t *m_valid, float score_weight,
                                        int max_iter, int proj_iter, float lr, int is_test, float *full_outmask,
                                        float *match_score, float *det_score, float *sim_out, float *R_out, float *Rb_out,
                                        int32_t *iters_out, void *workspace, size_t workspace_bytes, dmm_stream_t stream) {
    if (B < 0 || N < 0 || M < 0 || HW < 0 || D < 0) return DMM_ERR_BAD_ARG;
    if (B == 0 || M == 0) return DMM_OK;
    if (N == 0) return DMM_ERR_BAD_ARG;
    if (!masks_p || !packed_p || !masks_t || !feat_p || !feat_t || !score_p || !full_outmask || !match_score ||
        !det_score || !workspace)
        return DMM_ERR_BAD_ARG;
    const int Pp = N > M ? N : M + 1;
    if (M > DMM_MAX_TEMPLATES || Pp > DMM_MAX_PROPOSALS) return DMM_ERR_UNSUPPORTED;
    if (mask_dtype != DMM_F32 && mask_dtype != DMM_F16 && mask_dtype != DMM_BF16) return DMM_ERR_BAD_ARG;
    if (st_b != (int64_t)M * st_m) return DMM_ERR_UNSUPPORTED;          // templates: one plane stride over the batch
    dmm::Workspace w = dmm::carve(workspace, B, N, M, D);
    const int64_t wd = dmm_pack_words(HW);
    if (workspace_bytes < w.bytes + packed_t_bytes(B, M, HW)) return DMM_ERR_WORKSPACE;
    uint64_t *packed_t = (uint64_t *)((char *)workspace + w.bytes);
    float *sim = sim_out ? sim_out : w.sim;
    float *Rb = Rb_out ? Rb_out : w.Rb;
    int rc = dmm_pack_masks(masks_t, mask_dtype, (int64_t)B * M, HW, st_m, packed_t, wd, stream);
    if (rc != DMM_OK) return rc;
    // Feature similarity of all frames in the one-launch kernel, which also clears the count tables: every frame in the
    // summation order of ITS live proposal count (n_valid), template rows past m_valid are computed and never read (the
    // solver masks them; the order over D does not depend on the template count) -- bit identical to the ragged
    // three-launch form on every live entry.
    const bool force_tile = dmm::opt(DMM_OPT_COSINE_KERNEL) == 1;
    rc = force_tile ? DMM_ERR_UNSUPPORTED
                    : dmm::cosine_lanes_launch(feat_t, feat_p, B, N, M, D, w.cosv, (hipStream_t)stream, w.inter,
                                               (int64_t)B * M * N + (int64_t)B * N + (int64_t)B * M, n_valid);
    if (rc == DMM_OK) {
        rc = dmm::iou_counts_prezeroed(packed_p, packed_t, DMM_PACKED1, B, N, M, HW, pk_b, pk_n, (int64_t)M * wd, wd,
                                       n_valid, m_valid, w.inter, w.area_p, w.area_t, stream);
        if (rc != DMM_OK) return rc;
    } else if (rc != DMM_ERR_UNSUPPORTED) {
        return rc;
    } else {
        rc = dmm_iou_counts(packed_p, packed_t, DMM_PACKED1, B, N, M, HW, pk_b, pk_n, (int64_t)M * wd, wd, n_valid,
                            m_valid, w.inter, w.area_p, w.area_t, stream);
        if (rc != DMM_OK) return rc;
        rc = dmm_feature_normalize_f32(feat_p, (int64_t)B * N, D, w.featn_p, nullptr, stream);
        if (rc != DMM_OK) return rc;
        rc = dmm_feature_normalize_f32(feat_t, (int64_t)B * M, D, w.featn_t, nullptr, stream);
        if (rc != DMM_OK) return rc;
        rc = dmm_cosine_f32(w.featn_t, w.featn_p, B, N, M, D, n_valid, m_valid, w.cosv, stream);
        if (rc != DMM_OK) return rc;
    }
    rc = dmm_relax_match_f32(w.cosv, w.inter, w.area_p, w.area_t, score_p, B, N, M, n_valid, m_valid, score_weight,
                             max_iter, proj_iter, lr, is_test, sim, R_out, Rb, match_score, det_score, iters_out,
                             nullptr, stream);
    if (rc != DMM_OK) return rc;
    // train mode keeps every R > 0.01: the rows share planes -> the union of the supports is streamed once
    if (!is_test)
        return dmm_mask_mix_shared_to(Rb, masks_p, mask_dtype, B, N, M, Pp, HW, sp_b, sp_n, n_valid, m_valid, full_outmask,
                                      DMM_F32, (int64_t)M * HW, HW, stream);
    return dmm_mask_mix(Rb, masks_p, mask_dtype, B, N, M, Pp, HW, sp_b, sp_n, n_valid, m_valid, full_outmask,
                        (int64_t)M * HW, HW, stream);
}

// (5c) Cost + assignment of the fixed-slot frame step, BOTH sides of the cost pass on 1-bit planes and no mix: the
// proposals' words from dmm_paste_kept_f32, the templates' words from the previous frame's dmm_step_finish_f32.
// cosine (dense, also clears the count tables) -> counts on the words -> solver; Rb / scores / iters out.
extern "C" int dmm_match_solve_packed(const uint64_t *packed_p, const uint64_t *packed_t, const float *feat_p,
                                      const float *feat_t, const float *score_p, int B, int N, int M, int HW, int D,
                                      const int32_t *n_valid, const int32_t *m_valid, float score_weight, int max_iter,
                                      int proj_iter, float lr, int is_test, float *Rb_out, float *match_score,
                                      float *det_score, float *sim_out, float *R_out, int32_t *iters_out, void *workspace,
                                      size_t workspace_bytes, dmm_stream_t stream) {
    if (B < 0 || N < 0 || M < 0 || HW < 0 || D < 0) return DMM_ERR_BAD_ARG;
    if (B == 0 || M == 0) return DMM_OK;
    if (N == 0) return DMM_ERR_BAD_ARG;
    if (!packed_p || !packed_t || !feat_p || !feat_t || !score_p || !Rb_out || !match_score || !det_score || !workspace)
        return DMM_ERR_BAD_ARG;
    const int Pp = N > M ? N : M + 1;
    if (M > DMM_MAX_TEMPLATES || Pp > DMM_MAX_PROPOSALS) return DMM_ERR_UNSUPPORTED;
    dmm::Workspace w = dmm::carve(workspace, B, N, M, D);
    if (workspace_bytes < w.bytes) return DMM_ERR_WORKSPACE;
    const int64_t wd = dmm_pack_words(HW);
    float *sim = sim_out ? sim_out : w.sim;
    const bool force_tile = dmm::opt(DMM_OPT_COSINE_KERNEL) == 1;
    int rc = force_tile ? DMM_ERR_UNSUPPORTED
                        : dmm::cosine_lanes_launch(feat_t, feat_p, B, N, M, D, w.cosv, (hipStream_t)stream, w.inter,
                                                   (int64_t)B * M * N + (int64_t)B * N + (int64_t)B * M, n_valid);
    if (rc == DMM_OK) {
        rc = dmm::iou_counts_prezeroed(packed_p, packed_t, DMM_PACKED1, B, N, M, HW, (int64_t)N * wd, wd, (int64_t)M * wd,
                                       wd, n_valid, m_valid, w.inter, w.area_p, w.area_t, stream);
        if (rc != DMM_OK) return rc;
    } else if (rc != DMM_ERR_UNSUPPORTED) {
        return rc;
    } else {
        rc = dmm_iou_counts(packed_p, packed_t, DMM_PACKED1, B, N, M, HW, (int64_t)N * wd, wd, (int64_t)M * wd, wd, n_valid,
                            m_valid, w.inter, w.area_p, w.area_t, stream);
        if (rc != DMM_OK) return rc;
        rc = dmm_feature_normalize_f32(feat_p, (int64_t)B * N, D, w.featn_p, nullptr, stream);
        if (rc != DMM_OK) return rc;
        rc = dmm_feature_normalize_f32(feat_t, (int64_t)B * M, D, w.featn_t, nullptr, stream);
        if (rc != DMM_OK) return rc;
        rc = dmm_cosine_f32(w.featn_t, w.featn_p, B, N, M, D, n_valid, m_valid, w.cosv, stream);
        if (rc != DMM_OK) return rc;
    }
    return dmm_relax_match_f32(w.cosv, w.inter, w.area_p, w.area_t, score_p, B, N, M, n_valid, m_valid, score_weight,
                               max_iter, proj_iter, lr, is_test, sim, R_out, Rb_out, match_score, det_score, iters_out,
                               nullptr, stream);
}
