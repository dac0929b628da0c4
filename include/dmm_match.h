/*
 * dmm_match.h -- C ABI of libdmm_match.so: the MI355X (gfx950) implementation of DMM-Net's
 * differentiable mask-matching layer.
 *
 * Drop-in boundary.  The reference (ZENGXH/DMM_Net) is pure Python; its matching layer is
 *   dmm/modules/match_model.py:13-152   class MatchModel (forward :24-47)
 *   dmm/utils/match_helper.py:9-64      pairwise mask IoU, cosine, matching loss
 *   dmm/modules/submodules/relax_match.py:9-105   relax_matching (PGD + Dykstra projections)
 * and it reaches native code only through torch ops.  The entry points below are what a
 * ctypes / cpp_extension binding of that layer calls instead of those torch ops; each one
 * names the reference lines it replaces.  See INTEGRATION.md for the reference-side stub.
 *
 * Conventions
 *  - plain C: raw DEVICE pointers, sizes, element strides, a hipStream_t passed as void*;
 *    no torch types, no exceptions; every function returns a dmm_status (0 = ok);
 *  - nothing is allocated: outputs and workspace are caller provided (dmm_workspace_bytes);
 *  - everything is enqueued on `stream` and is hipGraph-capturable (no host syncs);
 *  - batched: B independent frames per call.  B = 1 reproduces one reference call;
 *    N = proposals (reference "P"), M = templates (reference "O"), Pp = max(N, M+1) is the
 *    padded solver width (match_model.py:109-113), HW = H*W pixels of one mask plane;
 *  - ragged batches: n_valid[B] / m_valid[B] (device int32, may be NULL) give the number of
 *    live proposals / templates of each frame (<= N, <= M); tables keep the N/M/Pp strides;
 *    frames with m_valid == 0 or n_valid == 0 produce zeros (dmm_model.py:118-122);
 *  - a mask plane is H*W contiguous elements; planes / frames may be strided (in elements);
 *  - integer results are bit exact w.r.t. the reference; fp32 results follow the reference's
 *    op order with fp contraction disabled (sums use a fixed tree order).
 */
#ifndef DMM_MATCH_H
#define DMM_MATCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DMM_ABI_VERSION 2

typedef enum dmm_status {
    DMM_OK = 0,
    DMM_ERR_BAD_ARG = 1,      /* null pointer / negative size / inconsistent shape */
    DMM_ERR_UNSUPPORTED = 2,  /* shape outside the compiled kernel envelope          */
    DMM_ERR_LAUNCH = 3,       /* HIP reported an error; see dmm_last_hip_error()     */
    DMM_ERR_WORKSPACE = 4     /* workspace too small                                 */
} dmm_status;

typedef enum dmm_dtype {
    DMM_F32 = 0, DMM_F16 = 1, DMM_BF16 = 2,
    DMM_PACKED1 = 3   /* 1 bit per pixel, already thresholded; uint64 words in the library's ballot layout (see (1c)) */
} dmm_dtype;

typedef void *dmm_stream_t; /* hipStream_t */

#define DMM_API __attribute__((visibility("default")))

/* Envelope of the FAST kernels: the solver keeps a frame's table in registers (M rows, Pp = max(N, M+1) columns).
 * DMM-Net's configurations sit well inside it (<= 100 proposals, a handful of objects).  The reference itself is
 * unbounded (relax_match.py:36-105), and so is the layer here, FORWARD AND BACKWARD: dmm_match_forward (5), dmm_cosine_f32
 * (2), dmm_mask_mix* (4), dmm_mask_mix_bwd, dmm_relax_match_bwd_f32 (its workspace holds the general kernel's state: ask
 * dmm_relax_bwd_workspace_bytes) and dmm_feature_sim_bwd_f32 take ANY N and M -- beyond the envelope through general kernels
 * (same operations in the same order, bit identical to the reference's CPU path in the forward; written for correctness,
 * not speed), dmm_iou_counts_* tiles any N x M.  The granular solver entries dmm_relax_match_f32 / _f16s / dmm_relax_solve_*
 * (no argument to hold the general solver's state: use dmm_relax_match_any_f32 (3d)), the 1-bit forms (5b) / (5c) and the
 * frame-step entries keep the envelope and answer DMM_ERR_UNSUPPORTED outside it. */
#define DMM_MAX_TEMPLATES 32  /* M  */
#define DMM_MAX_PROPOSALS 256 /* Pp */

/* Frame-stride sentinel of the entries that take (masks_p, sp_b): with sp_b == DMM_FRAME_TABLE, masks_p is a DEVICE array of
 * B pointers to each frame's first proposal plane (what the *_frames entries of (1d) / (4c) pass on) instead of the base
 * of B equally spaced frames. */
#define DMM_FRAME_TABLE INT64_MIN

/* ---------------------------------------------------------------------------------------------
 * (0) Dispatch options.  Where an entry point has two kernels (or a tuning value worth A/B-ing) the choice is a
 * process-wide integer set THROUGH THIS ABI.  The library never reads the environment: a stray variable cannot change what
 * production dispatches.  Defaults are the measured best; NO option changes a result (the parity tests pin each
 * alternative and compare bit for bit).  dmm_set_option answers DMM_ERR_BAD_ARG for an unknown option or a value outside
 * its range; dmm_get_option returns INT_MIN for an unknown option; dmm_reset_options restores every default.
 * Options are read at launch time: set them before the calls they should affect (they are not captured by value in
 * a HIP graph -- a captured launch keeps the choice it was captured with).
 * ------------------------------------------------------------------------------------------- */
typedef enum dmm_option {
    DMM_OPT_COST_KERNEL = 0,        /* dmm_iou_counts*: -1 by shape (default), 0 register tiles, 1 template lanes          */
    DMM_OPT_COST_TINY_FRAMES = 1,   /* largest B that takes the short-chunk instantiation of the register-tile kernel (8) */
    DMM_OPT_SOLVER_KERNEL = 2,      /* dmm_relax_*: -1 by shape (default), 0 thread per column, 1 row split               */
    DMM_OPT_FORCE_WIDE = 3,         /* 1: the any-size kernels also inside the fast envelope (0)                          */
    DMM_OPT_COSINE_KERNEL = 4,      /* fused feature similarity: 0 D over the lanes (default), 1 one thread per output    */
    DMM_OPT_COST_WGS = 5,           /* workgroup targets of the count kernels: register tiles (8192),                     */
    DMM_OPT_COST_SMALL_WGS = 6,     /*   sub-tiling threshold of small launches (512),                                     */
    DMM_OPT_COST_TL_WGS = 7,        /*   template lanes (512)                                                              */
    DMM_OPT_COST_XCD = 8,           /* XCD-aware workgroup -> (frame, range) mapping of the count kernels (1)              */
    DMM_OPT_MIX_XCD = 9,            /*   ... of the mix kernels, bits: 1 row kernel, 2 union kernel, 4 union backward (3)  */
    DMM_OPT_MIX_WGS = 10,           /* mix kernel: workgroup target (320000),                                              */
    DMM_OPT_MIX_STEPQ = 11,         /*   steps per workgroup quantum (2),                                                  */
    DMM_OPT_MIX_ALIGN = 12,         /*   store alignment in bytes: 16 / 32 / 64 / 128 (128),                               */
    DMM_OPT_MIX_NT = 13,            /*   non-temporal loads (bit 0) / stores (bit 1) (3)                                   */
    DMM_OPT_SOLVER_HELPER_MAX = 14, /* largest B whose one-wave solver launches carry the cost-norm helper wave (512)      */
    DMM_OPT_NMS_WAVE = 15,          /* dmm_nms_slots_f32: 1 the one-workgroup kernel for <= 64 boxes (default), 0 general  */
    DMM_OPT_COS_ROWS_MIN_N = 16,    /* dmm_cosine_f32: from which N the row-blocked form is used (65)                      */
    DMM_OPT_GEMM_TUNE = 17,         /* dmm_conv1x1_bf16: hipBLASLt heuristic candidates timed per new shape (1 = none)     */
    DMM_OPT_PACK_VARIANT = 18,      /* dmm_pack_masks: 4 = 128-block segments x 8 loads (default), 0 = 256 x 4             */
    DMM_OPT_SMALL_FUSED = 19,       /* dmm_match_forward, a handful of dense frames: 1 = the feature similarity rides in the
                                       count launch (similarity workgroups beside count workgroups; default), 0 = two launches */
    DMM_OPT_MIX_SHARED = 20,        /* mix / mix backward: -1 by entry point (default: dmm_mask_mix_shared_* and the backward
                                       stream the union of the rows' planes once), 0 row kernels always, 1 union kernels always */
    DMM_OPT_MIX_SHARED_STEPS = 21,  /* union kernels: 4 KiB steps of every plane per workgroup (1)                         */
    DMM_OPT_FEAT_BWD_FRAME = 22,    /* dmm_feature_sim_bwd_f32: -1 by batch size (default: one WAVE per feature row up to 256 frames,
                                       one workgroup per frame beyond), 2 / 1 / 0: wave per row / per frame / workgroup per row  */
    DMM_OPT_MIX_SHARED_LOCKSTEP = 23, /* union kernels: 1 = one workgroup barrier per group of 8 planes keeps the four waves (four
                                       neighbouring 1 KiB pieces of every plane) in step (default: the lines the pieces share are
                                       then asked for at the same time, -2 % traffic), 0 = free-running waves                  */
    DMM_OPT_COUNT = 24
} dmm_option;
DMM_API int dmm_set_option(int option, int value);
DMM_API int dmm_get_option(int option);
DMM_API int dmm_reset_options(void);

DMM_API int dmm_abi_version(void);
DMM_API const char *dmm_status_string(int status);
DMM_API int dmm_last_hip_error(void);         /* hipError_t of the last DMM_ERR_LAUNCH on this thread */
DMM_API const char *dmm_build_info(void);     /* "gfx950 ..." */
/* Diagnostic: kernels this library has enqueued in this process so far (every launch, the table-clearing ones included;
 * a replayed HIP graph is not counted again).  bench.py reports launches per call from differences of it. */
DMM_API long long dmm_launch_count(void);

/* ---------------------------------------------------------------------------------------------
 * (1) Pairwise binary-mask intersection / area tables.
 * Replaces compute_iou_binary_mask_2D over the expanded [O*P, HW] tensors
 * (match_helper.py:9-28 as called from match_model.py:83-89, and from
 * compute_matching_loss match_helper.py:34-43 with masks_t = targets).
 *   a = x > 0.5 (strict);  inter[b,m,n] = |T_m & P_n|;  area_p[b,n] = |P_n|;  area_t[b,m] = |T_m|
 * (union = area_p + area_t - inter).  Outputs are int32 and are overwritten.
 * masks_p: N planes per frame, plane stride sp_n, frame stride sp_b (elements);
 * masks_t: M planes per frame, strides st_m / st_b.
 * ------------------------------------------------------------------------------------------- */
DMM_API int dmm_iou_counts(const void *masks_p, const void *masks_t, int dtype, int B, int N, int M, int HW,
                   int64_t sp_b, int64_t sp_n, int64_t st_b, int64_t st_m,
                   const int32_t *n_valid, const int32_t *m_valid,
                   int32_t *inter /*[B,M,N]*/, int32_t *area_p /*[B,N]*/, int32_t *area_t /*[B,M]*/,
                   dmm_stream_t stream);

/* (1c) Bit-packed planes.  dmm_pack_masks thresholds (x > 0.5) and packs `planes` mask planes of HW pixels into
 * 4*ceil(HW/256) uint64 words each (ballot layout: bit l of word 4q+k = pixel 256q + 4l + k; pad bits 0);
 * dmm_pack_words(HW) returns that word count.  dmm_iou_counts / dmm_iou_counts_dual accept dtype DMM_PACKED1:
 * masks_* then point to such words and all strides are in WORDS.  Packing the proposal side once (or having
 * dmm_paste_masks_f32 emit it) shrinks the bytes of the cost pass 32x for those planes; the integer tables are
 * identical to the fp32 path by construction. */
DMM_API int64_t dmm_pack_words(int HW);
DMM_API int dmm_pack_masks(const void *masks, int dtype, int64_t planes, int HW, int64_t plane_stride,
                           uint64_t *packed, int64_t packed_stride, dmm_stream_t stream);

/* (1b) Training: the same pass also intersects the proposals with a SECOND template set of M planes per
 * frame -- the ground-truth targets of compute_matching_loss (match_helper.py:34-43) -- so the proposal
 * planes (the bulk of the bytes) are streamed once instead of twice.  inter2 [B,M,N], area_t2 [B,M]. */
DMM_API int dmm_iou_counts_dual(const void *masks_p, const void *masks_t, const void *masks_t2, int dtype, int B,
                                int N, int M, int HW, int64_t sp_b, int64_t sp_n, int64_t st_b, int64_t st_m,
                                int64_t st2_b, int64_t st2_m, const int32_t *n_valid, const int32_t *m_valid,
                                int32_t *inter, int32_t *area_p, int32_t *area_t, int32_t *inter2, int32_t *area_t2,
                                dmm_stream_t stream);

/* (1d) / (4c) PER-FRAME POINTER TABLES for the proposal planes.  The reference's driver holds the proposal masks
 * of a step as one tensor PER VIDEO (`prop_m[bid] = proposals[bid].get_field('mask').squeeze(1)`,
 * dmm/modules/dmm_model.py:58 / :111) and loops over them (:62 / :115).  The *_frames entry points take a DEVICE
 * array of B pointers -- masks_p_frames[b] = first plane of frame b, N_b planes at stride sp_n -- instead of
 * (masks_p, sp_b), so all videos go through one launch WITHOUT first being copied into a [B,Nmax,H,W] batch
 * (that copy moved 2 x 13 MB per video in front of a 15.6 MB cost pass).  Everything else as in (1), (1b), (4), (4b). */
DMM_API int dmm_iou_counts_frames(const void *const *masks_p_frames, const void *masks_t, int dtype, int B, int N,
                                  int M, int HW, int64_t sp_n, int64_t st_b, int64_t st_m, const int32_t *n_valid,
                                  const int32_t *m_valid, int32_t *inter, int32_t *area_p, int32_t *area_t,
                                  dmm_stream_t stream);
DMM_API int dmm_iou_counts_dual_frames(const void *const *masks_p_frames, const void *masks_t, const void *masks_t2,
                                       int dtype, int B, int N, int M, int HW, int64_t sp_n, int64_t st_b,
                                       int64_t st_m, int64_t st2_b, int64_t st2_m, const int32_t *n_valid,
                                       const int32_t *m_valid, int32_t *inter, int32_t *area_p, int32_t *area_t,
                                       int32_t *inter2, int32_t *area_t2, dmm_stream_t stream);
DMM_API int dmm_mask_mix_frames(const float *Rb, const void *const *masks_p_frames, int dtype, int B, int N, int M,
                                int Pp, int HW, int64_t sp_n, const int32_t *n_valid, const int32_t *m_valid,
                                float *out, int64_t so_b, int64_t so_m, dmm_stream_t stream);
DMM_API int dmm_mask_mix_bwd_frames(const float *Rb, const void *const *masks_p_frames, int dtype, const float *dout,
                                    int B, int N, int M, int Pp, int HW, int64_t sp_n, const int32_t *n_valid,
                                    const int32_t *m_valid, float *dRb, dmm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * (2) Row-normalise feature vectors: out[r,:] = in[r,:] / max(||in[r,:]||_2, 1e-8).
 * First half of F.cosine_similarity as get_cosine_score uses it (match_helper.py:51-64).
 * Also returns the clamped norms (needed by the backward).  rows = B*N or B*M.
 * ------------------------------------------------------------------------------------------- */
DMM_API int dmm_feature_normalize_f32(const float *in, int64_t rows, int D, float *out /*[rows,D]*/,
                              float *norms /*[rows] or NULL*/, dmm_stream_t stream);

/* (2c) get_cosine_score (match_helper.py:51-64) from the RAW features in one launch: (2) for both inputs + (2b), bit
 * identical to that sequence.  Dense batches only, N >= 2, D % 64 == 0.  D = 256 / 512 / 1024 (the product's 4 x
 * hidden_size) runs with the D axis spread over the lanes of a wave (dmm_cosine_lanes.hip); any other D with one thread
 * per output, the proposal columns tiled so that a tile plus the M template rows fit the LDS ((nt + M) * (D + 4) * 4
 * bytes <= 160 KB).  DMM_ERR_UNSUPPORTED when not even one column does (callers then run (2), (2), (2b)).  Used by (5). */
DMM_API int dmm_cosine_features_f32(const float *feat_t /*[B,M,D]*/, const float *feat_p /*[B,N,D]*/, int B, int N,
                                    int M, int D, float *cos_out /*[B,M,N]*/, dmm_stream_t stream);

/* (2d) Backward of the feature similarity: d/d feat_t, d/d feat_p of  sum(dsim * sim) + sum_b d_loss[b] * cost_loss[b],
 * i.e. torch autograd through get_cosine_score (match_helper.py:51-64), the (1 - score_weight) mix (match_model.py:90)
 * and compute_matching_loss's mse (match_helper.py:48).  featn_* / norm_* are the outputs of (2) saved by the forward;
 * gt [B,M,N] (greedy one-hot, no gradient), cos and d_loss [B] may be NULL together (inference-style loss-free call).
 * One launch; compared with the reference's autograd at 2e-5 relative to the largest gradient entry (tests). */
DMM_API int dmm_feature_sim_bwd_f32(const float *dsim /*[B,M,N]*/, const float *cos /*[B,M,N]*/, const float *gt,
                                    const float *d_loss, float score_weight, const float *feat_t, const float *feat_p,
                                    const float *featn_t, const float *featn_p, const float *norm_t, const float *norm_p,
                                    int B, int N, int M, int D, const int32_t *n_valid, const int32_t *m_valid,
                                    float *g_feat_t /*[B,M,D]*/, float *g_feat_p /*[B,N,D]*/, dmm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * (2b) Cosine table of normalised rows: cos[b,m,n] = <featn_t[b,m,:], featn_p[b,n,:]>
 * (second half of F.cosine_similarity, match_helper.py:59-63; == MatchModel's feature_sim for a
 * single template-feature entry, match_model.py:71-76).  featn_* come from (2).  M <= 32.
 * ------------------------------------------------------------------------------------------- */
DMM_API int dmm_cosine_f32(const float *featn_t /*[B,M,D]*/, const float *featn_p /*[B,N,D]*/, int B, int N, int M,
                           int D, const int32_t *n_valid, const int32_t *m_valid, float *cos_out /*[B,M,N]*/,
                           dmm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * (3) Similarity mix + relaxed assignment + scores, one frame per wave(-group), solver state
 * resident in registers for all iterations.  Replaces, per frame:
 *   iou = inter / (union + 1e-6);  sim = (1-w)*feature_sim + w*iou      match_model.py:89-90
 *   pad to [M, Pp];  C = -sim_pad                                      match_model.py:107-116
 *   relax_matching(C, max_iter, proj_iter, lr) and R = mean(X_list)
 *                                       relax_match.py:36-105, match_model.py:118-121
 *   logic = (R == rowmax) if is_test else (R > 0.01);  Rb = R*logic    match_model.py:124-130
 *   match_score = max_p clamp(R,0,1)*sim_pad;  det_score = sum_p score_p*Rb     :146-147
 * cos_in [B,M,N] is feature_sim (from (2b)).
 * Outputs: sim [B,M,N], R (may be NULL) / Rb [B,M,Pp], match_score / det_score [B,M],
 * iters (may be NULL) [B] = executed outer iterations (len(X_list)-1), X_final (may be NULL)
 * [B,M,Pp].  Requires M <= DMM_MAX_TEMPLATES and Pp <= DMM_MAX_PROPOSALS.
 * ------------------------------------------------------------------------------------------- */
DMM_API int dmm_relax_match_f32(const float *cos_in, const int32_t *inter, const int32_t *area_p,
                                const int32_t *area_t, const float *score_p /*[B,N]*/, int B, int N, int M,
                                const int32_t *n_valid, const int32_t *m_valid,
                                float score_weight, int max_iter, int proj_iter, float lr, int is_test,
                                float *sim_out, float *R_out, float *Rb_out,
                                float *match_score, float *det_score, int32_t *iters_out, float *X_final,
                                dmm_stream_t stream);

/* (3d) The same for ANY N and M (the reference's relax_matching is unbounded, relax_match.py:36-105): outside the fast
 * kernels' envelope the general solver keeps its state -- 9 tables of M x Pp floats per frame -- in `scratch`
 * (>= dmm_relax_any_scratch_bytes(B, N, M)); inside the envelope this is (3) and the scratch is not touched.  Same
 * outputs, bit identical to the reference's CPU path at every size. */
DMM_API size_t dmm_relax_any_scratch_bytes(int B, int N, int M);
DMM_API int dmm_relax_match_any_f32(const float *cos_in, const int32_t *inter, const int32_t *area_p,
                                    const int32_t *area_t, const float *score_p, int B, int N, int M,
                                    const int32_t *n_valid, const int32_t *m_valid, float score_weight, int max_iter,
                                    int proj_iter, float lr, int is_test, float *sim_out, float *R_out, float *Rb_out,
                                    float *match_score, float *det_score, int32_t *iters_out, float *X_final,
                                    void *scratch, size_t scratch_bytes, dmm_stream_t stream);

/* (3c) The same with the solver STATE in packed fp16 and every sum in fp32 -- BASELINE configs[4]'s "fp16 Sinkhorn with
 * fp32 accumulate".  A TOLERANCE mode, opt-in: the default (dmm_relax_match_f32) reproduces the reference bit for bit
 * including its exact-equality exits (relax_match.py:88-89, :96-98); here the iteration runs on fp16 iterates, so R is
 * within 1e-2 of the fp32 result (2.5e-3 measured on the config-2 / config-5 shapes) while both execute the same number of iterations, the row argmax is the same wherever
 * the fp32 decision is not a near tie, and the exits fire at the fp16 iteration's own fixed point.  What it buys: the
 * 20 x 200 problem needs <= 128 VGPRs (fp32: 256 + spills), four waves per SIMD, so the solver runs beside the streaming
 * cost / mix kernels instead of serialising with them. */
DMM_API int dmm_relax_match_f16s(const float *cos_in, const int32_t *inter, const int32_t *area_p,
                                 const int32_t *area_t, const float *score_p /*[B,N]*/, int B, int N, int M,
                                 const int32_t *n_valid, const int32_t *m_valid,
                                 float score_weight, int max_iter, int proj_iter, float lr, int is_test,
                                 float *sim_out, float *R_out, float *Rb_out,
                                 float *match_score, float *det_score, int32_t *iters_out, float *X_final,
                                 dmm_stream_t stream);

/* Solver only, on caller-provided cost matrices C [B,n,m] (relax_matching itself,
 * relax_match.py:36-105; with max_iter = 0 it is the greedy initialisation used by
 * compute_matching_loss, match_helper.py:44): X_final, R = mean(X_list), cost list [B,max_iter+1]
 * (may be NULL), iters [B].  rows_valid / cols_valid [B] (may be NULL) restrict frame b to its top-left
 * [rows_valid[b], cols_valid[b]] block (the rest of X_final / R is zero filled).
 * n <= DMM_MAX_TEMPLATES, m <= DMM_MAX_PROPOSALS. */
DMM_API int dmm_relax_solve_f32(const float *C, int B, int n, int m, const int32_t *rows_valid,
                                const int32_t *cols_valid, int max_iter, int proj_iter, float lr,
                                float *X_final, float *R_out, float *cost_out, int32_t *iters_out,
                                dmm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * (3b) Backward of (3) with respect to sim (the reference gets it from torch autograd through
 * relax_match.py:68-98 and match_model.py:121-147; the greedy init, the masks and the early exits
 * carry no gradient).  The kernel re-runs the forward solver from the saved `sim` (same code, hence
 * the same iterates and exits), tapes 1 relu bit per element + 1 column bit per sweep into
 * `workspace`, and walks the tape backwards.
 *   dRb [B,M,Pp]        upstream gradient of Rb (= dOut @ masks_p^T; logic is applied inside); may be NULL
 *   d_match_score, d_det_score [B,M]   upstream gradients of the two score vectors; may be NULL
 *   dsim_out [B,M,N]    gradient of the loss w.r.t. sim (feature_sim gets (1-w) * dsim)
 * workspace >= dmm_relax_bwd_workspace_bytes(...).  max_iter <= 1024.
 * ------------------------------------------------------------------------------------------- */
DMM_API size_t dmm_relax_bwd_workspace_bytes(int B, int N, int M, int max_iter, int proj_iter);

DMM_API int dmm_relax_match_bwd_f32(const float *sim, const float *score_p, int B, int N, int M,
                                    const int32_t *n_valid, const int32_t *m_valid,
                                    int max_iter, int proj_iter, float lr, int is_test,
                                    const float *dRb, const float *d_match_score, const float *d_det_score,
                                    float *dsim_out, void *workspace, size_t workspace_bytes,
                                    dmm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * (4) Assignment-weighted mask mix: full_outmask[b,m,:] = sum_n Rb[b,m,n] * masks_p[b,n,:]
 * (torch.mm at match_model.py:144; padded columns n >= N carry zero planes, :134-142).
 * Only planes with a non-zero weight are read (in test mode <= M planes per frame).
 * Rb is [B,M,Pp] with row stride Pp.  out: [B,M,HW] fp32, strides so_b / so_m (elements).
 * Rows m >= m_valid[b] are zero filled.
 * ------------------------------------------------------------------------------------------- */
DMM_API int dmm_mask_mix(const float *Rb, const void *masks_p, int dtype, int B, int N, int M, int Pp, int HW,
                 int64_t sp_b, int64_t sp_n, const int32_t *n_valid, const int32_t *m_valid,
                 float *out, int64_t so_b, int64_t so_m, dmm_stream_t stream);

/* (4a) Same with a choice of the output element type: DMM_F32, or the planes' own 16-bit type (dtype) -- BASELINE
 * config 5 keeps the masks in fp16: the matched masks are the next frame's templates (mask_last_occurence,
 * dmm_model.py:78-80), so they are written back in the storage type (the fp32 result rounded once, nearest-even).
 * out strides so_b / so_m are in OUTPUT elements. */
DMM_API int dmm_mask_mix_to(const float *Rb, const void *masks_p, int dtype, int B, int N, int M, int Pp, int HW,
                            int64_t sp_b, int64_t sp_n, const int32_t *n_valid, const int32_t *m_valid, void *out,
                            int out_dtype, int64_t so_b, int64_t so_m, dmm_stream_t stream);

/* (4d) dmm_mask_mix_to for weight tables whose ROWS SHARE PLANES -- train mode: logic = (R > 0.01) keeps many entries per
 * row (match_model.py:126-129), torch.mm then reads every plane once for all rows (:144).  One workgroup streams each
 * plane of the union of the rows' supports once and fans it into the rows; per row the accumulation is the one of
 * dmm_mask_mix_to (non-zero weights in ascending column order), so the two entries agree bit for bit.  M <= 32, N <= 256
 * take the union kernel, anything else the general one.  dmm_mask_mix_shared_frames: proposal planes as a device table of
 * per-frame base pointers (see (4c)). */
DMM_API int dmm_mask_mix_shared_to(const float *Rb, const void *masks_p, int dtype, int B, int N, int M, int Pp, int HW,
                                   int64_t sp_b, int64_t sp_n, const int32_t *n_valid, const int32_t *m_valid,
                                   void *out, int out_dtype, int64_t so_b, int64_t so_m, dmm_stream_t stream);
DMM_API int dmm_mask_mix_shared_frames(const float *Rb, const void *const *masks_p_frames, int dtype, int B, int N,
                                       int M, int Pp, int HW, int64_t sp_n, const int32_t *n_valid,
                                       const int32_t *m_valid, float *out, int64_t so_b, int64_t so_m,
                                       dmm_stream_t stream);

/* (4b) Backward of (4) w.r.t. Rb: dRb[b,m,n] = <dout[b,m,:], masks_p[b,n,:]> on the support of Rb (entries with
 * Rb == 0 were masked by the constant logic mask, match_model.py:124-130, and get 0).  dout: [B,M,HW] fp32
 * contiguous; dRb: [B,M,Pp] fp32, overwritten.  Only the selected planes are read. */
DMM_API int dmm_mask_mix_bwd(const float *Rb, const void *masks_p, int dtype, const float *dout, int B, int N, int M,
                             int Pp, int HW, int64_t sp_b, int64_t sp_n, const int32_t *n_valid,
                             const int32_t *m_valid, float *dRb, dmm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * (5) The whole forward of MatchModel for B frames (match_model.py:24-47, targets=None):
 * (1) -> (2) -> (3) -> (4) on `stream`, intermediates in `workspace`.
 * Outputs as in (3)/(4).  workspace >= dmm_workspace_bytes(B, N, M, D).  Any N, M: outside the fast kernels' envelope
 * (M > 32 or max(N, M+1) > 256) the workspace also holds the general solver's state (9 tables of M x Pp floats per
 * frame; dmm_workspace_bytes accounts for it) and the general kernels run inside this call.
 * ------------------------------------------------------------------------------------------- */
DMM_API size_t dmm_workspace_bytes(int B, int N, int M, int D);

DMM_API int dmm_match_forward(const void *masks_p, const void *masks_t, int mask_dtype,
                      const float *feat_p /*[B,N,D]*/, const float *feat_t /*[B,M,D]*/,
                      const float *score_p /*[B,N]*/, int B, int N, int M, int HW, int D,
                      int64_t sp_b, int64_t sp_n, int64_t st_b, int64_t st_m,
                      const int32_t *n_valid, const int32_t *m_valid,
                      float score_weight, int max_iter, int proj_iter, float lr, int is_test,
                      float *full_outmask /*[B,M,HW]*/, float *match_score /*[B,M]*/,
                      float *det_score /*[B,M]*/, float *sim_out /*[B,M,N] or NULL*/,
                      float *R_out /*[B,M,Pp] or NULL*/, float *Rb_out /*[B,M,Pp] or NULL*/,
                      int32_t *iters_out /*[B] or NULL*/,
                      void *workspace, size_t workspace_bytes, dmm_stream_t stream);

/* (5a') dmm_match_forward for a caller that keeps ONE workspace over a sequence of calls (the per-frame use) and carries
 * the library's note about what it left there: *ws_state in = the value the previous call on this workspace wrote (same
 * B, N, M, D; DMM_WS_UNKNOWN for a fresh or otherwise touched workspace), out = the state the work enqueued by this call
 * leaves behind.  With DMM_WS_TABLES_ZERO a handful of dense frames run without a clearing launch in front of the IoU
 * counts (the solver zeroes each count-table entry right after reading it); results are the same either way.  The state
 * describes the workspace AFTER the enqueued work: calls that share a workspace must be ordered on one stream, and a
 * captured call must be replayed with the workspace in the state it was captured with (it is, if nothing else touches
 * the workspace between replays: a call that starts from DMM_WS_TABLES_ZERO and returns it leaves what it found).
 * ws_state == NULL: exactly dmm_match_forward.
 * The CALLER owns this note: reset it to DMM_WS_UNKNOWN whenever anything but the previous dmm_match_forward_ws call with
 * the same (B, N, M, D) may have written the workspace -- another shape (the carving moves), another entry point, a
 * re-allocation, a second stream or thread sharing it.  A stale DMM_WS_TABLES_ZERO makes the counts accumulate onto old
 * tables: a silently wrong result the library cannot detect.  (dmm_net_amd/ops.py keeps one note per (device, stream,
 * shape) and drops it on every such event; the dispatch options of (0) are process-wide and not part of this state.) */
enum { DMM_WS_UNKNOWN = 0, DMM_WS_TABLES_ZERO = 1 };
DMM_API int dmm_match_forward_ws(const void *masks_p, const void *masks_t, int mask_dtype, const float *feat_p,
                                 const float *feat_t, const float *score_p, int B, int N, int M, int HW, int D,
                                 int64_t sp_b, int64_t sp_n, int64_t st_b, int64_t st_m, const int32_t *n_valid,
                                 const int32_t *m_valid, float score_weight, int max_iter, int proj_iter, float lr,
                                 int is_test, float *full_outmask, float *match_score, float *det_score, float *sim_out,
                                 float *R_out, float *Rb_out, int32_t *iters_out, void *workspace, size_t workspace_bytes,
                                 int *ws_state, dmm_stream_t stream);

/* (5b) The same forward with the proposal side of the cost pass on 1-bit planes the caller already holds
 * (dmm_paste_masks_f32 / dmm_paste_kept_f32 emit them next to the soft planes): packed_p [B,N,words] uint64, strides
 * pk_b / pk_n in WORDS.  The M template planes of every frame are packed inside (st_b == M * st_m required), the counts
 * run on the words -- identical integer tables from 1/32 of the proposal bytes -- and the mix reads the soft planes.
 * The feature similarity runs in the one-launch kernel for all frames, each in the summation order of ITS live proposal
 * count (the reference is called per frame; ATen's order over the [D, P] products depends on P), template rows past m_valid
 * are computed and never read.
 * This is the per-frame call of the evaluator's loop (dmm_model.py:75-77 for all videos of the step at once).
 * workspace >= dmm_workspace_bytes_packed(B, N, M, D, HW). */
DMM_API size_t dmm_workspace_bytes_packed(int B, int N, int M, int D, int HW);
DMM_API int dmm_match_forward_packed(const void *masks_p, const uint64_t *packed_p, const void *masks_t, int mask_dtype,
                                     const float *feat_p, const float *feat_t, const float *score_p, int B, int N, int M,
                                     int HW, int D, int64_t sp_b, int64_t sp_n, int64_t pk_b, int64_t pk_n, int64_t st_b,
                                     int64_t st_m, const int32_t *n_valid, const int32_t *m_valid, float score_weight,
                                     int max_iter, int proj_iter, float lr, int is_test, float *full_outmask,
                                     float *match_score, float *det_score, float *sim_out, float *R_out, float *Rb_out,
                                     int32_t *iters_out, void *workspace, size_t workspace_bytes, dmm_stream_t stream);

/* (5c) Cost + assignment of a fixed-slot frame step with BOTH sides of the cost pass on 1-bit planes and no mix (the
 * mix is part of (8c)): packed_p [B,N,words] from dmm_paste_kept_f32, packed_t [B,M,words] from the previous frame's
 * dmm_step_finish_f32 (or dmm_pack_masks for the first frame), both dense.  cosine -> counts -> solver
 * (match_model.py:49-130, :146-147); Rb [B,M,Pp], match_score / det_score [B,M], sim (may be NULL), R (may be NULL),
 * iters (may be NULL).  workspace >= dmm_workspace_bytes(B, N, M, D). */
DMM_API int dmm_match_solve_packed(const uint64_t *packed_p, const uint64_t *packed_t, const float *feat_p,
                                   const float *feat_t, const float *score_p, int B, int N, int M, int HW, int D,
                                   const int32_t *n_valid, const int32_t *m_valid, float score_weight, int max_iter,
                                   int proj_iter, float lr, int is_test, float *Rb_out, float *match_score,
                                   float *det_score, float *sim_out, float *R_out, int32_t *iters_out, void *workspace,
                                   size_t workspace_bytes, dmm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * (1e) The tail of compute_matching_loss (match_helper.py:43-48) on the device, after the counts of (1b):
 *   gt_iou = inter2 / (area_p + area_t2 - inter2 + 1e-6)  (compute_iou_binary_mask_2D on proposals x targets, :9-28, :43)
 *   gt     = relax_matching(-gt_iou, 0, 0, 0)[0]          (the greedy one-hot initialisation, relax_match.py:45-55; :44)
 *   loss   = F.mse_loss(feature_sim, gt)                   (:48; feature_sim = cos, BEFORE the IoU mix)
 * gt [B,M,N] (zeros outside a ragged frame's live [m_valid, n_valid] block), loss [B] (0 for a frame without live
 * templates or proposals, dmm_model.py:118-122).  One launch; N <= 8192, M <= 4096.
 * ------------------------------------------------------------------------------------------- */
DMM_API int dmm_matching_loss_f32(const int32_t *inter2 /*[B,M,N]*/, const int32_t *area_p /*[B,N]*/,
                                  const int32_t *area_t2 /*[B,M]*/, const float *cos /*[B,M,N]*/, int B, int N, int M,
                                  const int32_t *n_valid, const int32_t *m_valid, float *gt /*[B,M,N]*/,
                                  float *loss /*[B]*/, dmm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * (5d) / (5e) The TRAINING call of the layer -- MatchModel.forward with targets as the trainer issues it once per (video,
 * frame), dmm_model.py:130-132, and torch autograd back through it -- as ONE entry each way, so that a one-frame call
 * costs the host two library calls instead of twelve (the drop-in was host bound there).
 *
 * dmm_match_train_forward: (2c) feature similarity -> (1b) counts against templates AND targets in one pass over the
 *   proposal planes (targets == NULL: (1) and no loss) -> (3) -> (4d) [is_test: (4)] -> (1e).  Same kernels, same results
 *   bit for bit as the granular entries.  sp_b may be DMM_FRAME_TABLE.  targets: M planes per frame of the masks' dtype,
 *   strides sg_b / sg_m.  Outputs full_outmask [B,M,HW] fp32 contiguous, match_score / det_score [B,M], cost_loss [B]
 *   (NULL without targets), iters [B] (may be NULL); and what (5e) needs: cos, sim [B,M,N], Rb [B,M,Pp], gt [B,M,N]
 *   (NULL without targets).  Inside the fast kernels' envelope only (DMM_ERR_UNSUPPORTED otherwise: wider tables train
 *   through the granular entries).  workspace >= dmm_match_train_forward_workspace_bytes(B, N, M, D).
 * dmm_match_train_backward: d/d feat_t, d/d feat_p of
 *     sum(d_full * full_outmask) + sum(d_match_score * match_score) + sum(d_det_score * det_score) + sum(d_loss * cost_loss)
 *   = (2) on both feature sets (one launch; the forward keeps no normalised rows) -> (4b) -> (3b) -> (2d).  Any of d_full
 *   [B,M,HW], d_match_score, d_det_score [B,M] may be NULL (no gradient from that output); gt / d_loss / cos NULL together
 *   (no loss term).  Any N, M.  workspace >= dmm_match_train_backward_workspace_bytes(B, N, M, D, max_iter, proj_iter).
 * The solver's TAPE (optional, both entries): the reference's autograd keeps every intermediate of relax_matching
 *   (relax_match.py:68-98) for its backward; (3b) alone re-runs the solver to rebuild what it needs.  Given a caller block
 *   `tape` of dmm_match_train_tape_bytes(B, N, M, max_iter, proj_iter) bytes (0 = this table is not taped: wider than 64 solver
 *   columns) and iters_out != NULL, the forward's solver kernel records it there (R, and per projection sweep and column 8
 *   bytes of gate bits) and sets *taped = 1; handed back with that flag and the forward's iters, the backward walks the
 *   records instead of re-running the solver (same gradient: the same gates).  tape == NULL / taped == 0: as before.
 * ------------------------------------------------------------------------------------------- */
DMM_API size_t dmm_match_train_tape_bytes(int B, int N, int M, int max_iter, int proj_iter);
DMM_API size_t dmm_match_train_forward_workspace_bytes(int B, int N, int M, int D);
DMM_API int dmm_match_train_forward(const void *masks_p, const void *masks_t, const void *targets, int mask_dtype,
                                    const float *feat_p, const float *feat_t, const float *score_p, int B, int N, int M,
                                    int HW, int D, int64_t sp_b, int64_t sp_n, int64_t st_b, int64_t st_m, int64_t sg_b,
                                    int64_t sg_m, const int32_t *n_valid, const int32_t *m_valid, float score_weight,
                                    int max_iter, int proj_iter, float lr, int is_test, float *full_outmask,
                                    float *match_score, float *det_score, float *cost_loss, int32_t *iters_out,
                                    float *cos_out, float *sim_out, float *Rb_out, float *gt_out, void *workspace,
                                    size_t workspace_bytes, void *tape, size_t tape_bytes, int *taped /* host, may be NULL */,
                                    dmm_stream_t stream);
DMM_API size_t dmm_match_train_backward_workspace_bytes(int B, int N, int M, int D, int max_iter, int proj_iter);
DMM_API int dmm_match_train_backward(const void *masks_p, int mask_dtype, const float *feat_p, const float *feat_t,
                                     const float *score_p, const float *cos, const float *sim, const float *Rb,
                                     const float *gt, const float *d_full, const float *d_match_score,
                                     const float *d_det_score, const float *d_loss, int B, int N, int M, int HW, int D,
                                     int64_t sp_b, int64_t sp_n, const int32_t *n_valid, const int32_t *m_valid,
                                     float score_weight, int max_iter, int proj_iter, float lr, int is_test,
                                     float *g_feat_t /*[B,M,D]*/, float *g_feat_p /*[B,N,D]*/, void *workspace,
                                     size_t workspace_bytes, const void *tape, const int32_t *iters /*[B], the forward's*/,
                                     int taped, dmm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * (6) Fused 4-level ROIAlign + spatial mean: the reference's ROI feature extractor
 * (dmm/modules/feature_extractor.py:20-52: maskrcnn_benchmark legacy ROIAlign, 14x14 bins,
 * sampling_ratio 2, at the four scales on EVERY roi, then .mean(4).mean(3) -> [R, 4*C]).
 * feat[l]: [B, C, H[l], W[l]] contiguous NCHW (dtype fp32 / fp16 / bf16), rois: [R,5] fp32
 * (batch index, x1, y1, x2, y2) in image coordinates, scale[l] = 1/stride.  out: [R, 4*C] fp32,
 * out[r, l*C + c].  The backward accumulates (fp32 atomics) into dfeat[l] (same shapes, fp32,
 * zeroed by the caller); boxes receive no gradient (as in maskrcnn_benchmark).
 * H[l], W[l] <= 1024.  No upstream fixture exists for this third-party op; forward and gradients are pinned against an
 * independent differentiable formulation of the published per-bin definition (tests/golden G12) and the oracle.
 * dmm_roialign4_mean_nhwc_fwd: the same forward on channels-last features, feat[l] = [B, H[l], W[l], C] contiguous (what
 * the inference encoder produces): 16-byte lane loads of contiguous channels.  Needs C % (16 / element size) == 0 with
 * the quotient a power of two <= 64 or a multiple of 64, 16-byte aligned bases; otherwise DMM_ERR_UNSUPPORTED.
 * ------------------------------------------------------------------------------------------- */
DMM_API int dmm_roialign4_mean_nhwc_fwd(const void *const feat[4], int dtype, int B, int C, const int H[4],
                                        const int W[4], const float scale[4], const float *rois, int R, float *out,
                                        dmm_stream_t stream);
DMM_API int dmm_roialign4_mean_fwd(const void *const feat[4], int dtype, int B, int C, const int H[4], const int W[4],
                                   const float scale[4], const float *rois, int R, float *out, dmm_stream_t stream);
DMM_API int dmm_roialign4_mean_bwd(const float *dout, int B, int C, const int H[4], const int W[4],
                                   const float scale[4], const float *rois, int R, float *const dfeat[4],
                                   dmm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * (7) Proposal preprocessing (the step right before the path; host Python loops in the reference).
 * dmm_paste_masks_f32: paste_mask_in_image + binmask_to_box (dmm/utils/masker.py:110-173) for P proposals:
 *   prob [P,M,M] mask probabilities, boxes [P,4] xyxy -> planes [P, im_h*im_w] (plane_stride elements apart; the
 *   soft masks the matching layer consumes) and new_boxes [P,4] = tight box of (plane > thresh), or
 *   [0,0,im_h,im_w] when empty.  M + 2*padding <= 64.  Optionally also emits the DMM_PACKED1 form of the planes.
 * dmm_nms_f32: NMS + top-k of filter_results (dmm/utils/boxlist_ops.py:15-29, maskrcnn_benchmark nms semantics:
 *   descending score, legacy +1 areas, IoU > thresh suppresses) per image: boxes [sum n,4], scores [sum n],
 *   offsets [images+1] (device int32) -> keep[offsets[i] ...] = kept local indices in score order,
 *   keep_count[i].  max_per_image (<= 1024) bounds n.
 * ------------------------------------------------------------------------------------------- */
DMM_API int dmm_paste_masks_f32(const float *prob, int P, int M, const float *boxes, int im_h, int im_w, float thresh,
                                int padding, float *planes, int64_t plane_stride, float *new_boxes,
                                uint64_t *packed /* NULL or [P, dmm_pack_words(im_h*im_w)]: (plane > 0.5) bits */,
                                dmm_stream_t stream);
DMM_API int dmm_nms_f32(const float *boxes, const float *scores, const int32_t *offsets, int images, int max_per_image,
                        float thresh, int max_keep, int32_t *keep, int32_t *keep_count, dmm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * (7b) The same preprocessing in TWO PHASES on FIXED SLOTS, with no host round trip: what the evaluator does per frame
 * before the layer (dmm/modules/model_encoder.py:115-134: Masker paste of every raw proposal, then filter_results =
 * NMS + top-k, dmm/utils/boxlist_ops.py:15-29, then BoxList indexing of the kept ones) without ever writing the planes
 * of proposals NMS drops and without gathering the kept ones into a new tensor.
 *   raw inputs: prob [.., images, R, M, M], boxes [.., images, R, 4] xyxy, scores [.., images, R], counts [.., images]
 *   (NULL = R each) -- the leading axis is the FRAME of a clip-resident buffer, selected by the DEVICE scalar *step
 *   (NULL = frame 0), so a captured graph can replay the frame step of every frame of a clip (see (8b));
 * dmm_proposal_boxes_f32: tight [images, R, 4] = box of (pasted value > thresh) of every raw proposal, [0,0,im_h,im_w]
 *   when nothing passes (masker.py:164), zeros for empty raw slots -- the values are evaluated, no plane is written;
 * dmm_nms_slots_f32: NMS(thresh) + top-K on the tight boxes -> keep [images, K] raw indices in descending score order,
 *   keep_count [images] (stays on the device: it is the n_valid of the matching entry points); R <= 1024;
 * dmm_paste_kept_f32: slot (i, k), k < keep_count[i], receives raw proposal keep[i, k]: its soft plane
 *   planes[(i*K + k) * plane_stride ..] (NULL = do not write soft planes, see (8c)), its 1-bit plane packed [images, K,
 *   dmm_pack_words] (may be NULL), kept_boxes
 *   [images, K, 4] (the tight box), kept_scores [images, K] and the ROIAlign row rois [images*K, 5] = (img_base[*step]
 *   + i, tight box) (each may be NULL; img_base NULL = 0).  Dead slots: plane untouched, score 0, box 0, roi image
 *   index -1 (dmm_roialign4_mean_fwd writes a zero feature row for it).
 * Bit identical to dmm_paste_masks_f32 + dmm_nms_f32 + gather (same device function evaluates every pasted value).
 * ------------------------------------------------------------------------------------------- */
DMM_API int dmm_proposal_boxes_f32(const float *prob, const float *boxes, const int32_t *counts, int images, int R, int M,
                                   int im_h, int im_w, float thresh, int padding, const int32_t *step, float *tight,
                                   dmm_stream_t stream);
DMM_API int dmm_nms_slots_f32(const float *tight, const float *scores, const int32_t *counts, int images, int R,
                              float thresh, int K, const int32_t *step, int32_t *keep, int32_t *keep_count,
                              dmm_stream_t stream);
DMM_API int dmm_paste_kept_f32(const float *prob, const float *boxes, const float *scores, const float *tight,
                               const int32_t *keep, const int32_t *keep_count, int images, int R, int M, int K, int im_h,
                               int im_w, int padding, const int32_t *step, const int32_t *img_base, float *planes,
                               int64_t plane_stride, uint64_t *packed, float *kept_boxes, float *kept_scores, float *rois,
                               dmm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * (8) Frame-loop reductions (the step right after the path).
 * dmm_mask_boxes_f32: ohw_mask2boxlist (dmm/utils/utils.py:179-210, binmask_to_bbox_xyxy_pt :114-143) for R
 *   planes [R, H*W] (plane_stride elements apart): boxes [R,4] = tight xyxy box of (plane > thresh), or
 *   [0,0,W-1,H-1] when no pixel passes; valid [R] = 1 iff the plane has a pixel > 0 (reference: plane sum > 0).
 * dmm_merge_labels_f32: the per-frame label map of the evaluator (dmm/modules/evaluator.py:134-139):
 *   labels[b,x] = argmax([1 - max_o m[b,o,x], m[b,0,x], ..., m[b,O_b-1,x]]), first maximum wins (0 = background).
 *   masks [B,O,HW] with element strides; o_valid [B] int32 (NULL = O) live-template prefix per video; O <= 255.
 * ------------------------------------------------------------------------------------------- */
DMM_API int dmm_mask_boxes_f32(const float *masks, int R, int H, int W, int64_t plane_stride, float thresh,
                               float *boxes, int32_t *valid, dmm_stream_t stream);
DMM_API int dmm_merge_labels_f32(const float *masks, int B, int O, int HW, int64_t stride_b, int64_t stride_o,
                                 const int32_t *o_valid, uint8_t *labels, dmm_stream_t stream);

/* (8b) Device-resident frame cursor of the evaluator's loop (dmm/modules/evaluator.py:63-213: `for t in range(T)` with
 * every per-frame argument rebuilt on the host).  With the clip's raw proposals resident on the device ((7b)) and the
 * frame index in a device scalar, ONE captured HIP graph replays every frame step of the clip without host input:
 * dmm_step_select_i32: out[i] = table[*step * n + i] (a row of a per-clip table -- live template counts, commit flags
 *   per video -- copied to a fixed address); dmm_step_advance: *step += 1 (the graph's last node).
 * dmm_commit_masks_f32: out_mask_last of the per-video driver (dmm/modules/dmm_model.py:66-69 / :78-80): hist[b] =
 *   full[b] ([per_video] floats each) where commit[b] != 0; a skipped video (no live template, 'extra' frame) keeps its
 *   template planes. */
/* (8c) Frame-step epilogue on fixed slots: the mask mix (match_model.py:134-144), out_mask_last (dmm_model.py:66-69 /
 * :78-80) and the label map (evaluator.py:134-139) of every video of the step in ONE pass over the pixels, pasting the
 * few selected proposals ON THE FLY from their raw probabilities (so dmm_paste_kept_f32 may run with planes = NULL: the
 * soft planes of a frame, 10x the bytes of everything else in the step, are never written):
 *   full[b,m,:]  = sum_n Rb[b,m,n] * paste(raw proposal keep[b,n])   for m < m_valid[b], zeros otherwise (bit identical
 *                  to dmm_paste_kept_f32 + dmm_mask_mix: same paste arithmetic, same fma order);
 *   hist[b]      = full[b] where commit[b] != 0 (a skipped video keeps its template planes);
 *   packed_hist  = 1-bit planes of the new hist[b] (what the next frame's (5c) counts on; may be NULL);
 *   labels[b,x]  = as dmm_merge_labels_f32 over the first o_valid[b] rows of full (may be NULL).
 * Rb [B,M,Pp]; raw prob / boxes as in (7b) (clip resident, frame *step); keep / keep_count from dmm_nms_slots_f32.
 * M <= 8 rows, mask size + 2 padding <= 32; otherwise DMM_ERR_UNSUPPORTED (callers then paste the planes and use (4)). */
DMM_API int dmm_step_finish_f32(const float *Rb, int Pp, const float *prob, const float *boxes, const int32_t *keep,
                                const int32_t *keep_count, int B, int R, int Mm, int K, int M, int im_h, int im_w,
                                int padding, const int32_t *step, const int32_t *m_valid, const int32_t *commit,
                                const int32_t *o_valid, float *full, float *hist, uint64_t *packed_hist, uint8_t *labels,
                                dmm_stream_t stream);
DMM_API int dmm_step_select_i32(const int32_t *table, const int32_t *step, int n, int32_t *out, dmm_stream_t stream);
DMM_API int dmm_step_advance(int32_t *step, dmm_stream_t stream);
DMM_API int dmm_commit_masks_f32(const float *full, float *hist, const int32_t *commit, int B, int64_t per_video,
                                 dmm_stream_t stream);

/* dmm_ragged_pad: the batching step of the per-video driver (the reference loops MatchModel over the videos,
 *   dmm/modules/dmm_model.py:62-82 / :115-139; here they run as one ragged launch): out[b, i, :] = src_table[b][i, :] for
 *   i < counts[b], zeros up to P_max.  src_table: B device pointers (device memory) to contiguous [counts[b], row_bytes]
 *   blocks; counts [B] int32 (device); row_bytes % 4 == 0; out [B, P_max, row_bytes]. */
DMM_API int dmm_ragged_pad(const void *const *src_table, const int32_t *counts, int B, int P_max, int64_t row_bytes,
                           void *out, dmm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * (9) Encoder epilogue (inference, channels-last bf16): x[r, c] = act(x[r, c] + bias[c] (+ residual[r, c])) in place,
 * one pass, fp32 arithmetic, one rounding.  What remains of conv -> BatchNorm -> ReLU (dmm/modules/base.py:43-54,
 * model_encoder.py:137-146) and of the residual tails of the torchvision blocks (dmm/modules/vision.py:6-38) once the
 * BatchNorm is folded into the contraction.  x, residual: [rows, C] bfloat16 (16-byte lane accesses when C % 8 == 0); bias: [C] fp32 or NULL.
 * ------------------------------------------------------------------------------------------- */
DMM_API int dmm_bias_act_bf16(void *x, const float *bias, const void *residual, int64_t rows, int C, int relu,
                              dmm_stream_t stream);

/* (9d) Stem tail of the channels-last inference encoder: y = maxpool3x3/s2/p1( relu(x + bias[c]) ) in ONE pass over the stem
 * convolution's output x [B, H, W, C] bf16 (C % 8 == 0) -> y [B, (H-1)/2+1, (W-1)/2+1, C]; bit identical to
 * dmm_bias_act_bf16 followed by the pool (bias, relu and rounding are monotone, so the window maximum is taken first).
 * Replaces bn1 -> relu -> maxpool of the torchvision bodies (dmm/modules/vision.py:11-21 forward) after BatchNorm folding. */
DMM_API int dmm_bias_relu_maxpool_bf16(const void *x, const float *bias, int B, int H, int W, int C, void *y,
                                       dmm_stream_t stream);

/* (9c) Patch matrix of a 3x3 / padding 1 / stride 1|2 convolution on a channels-last bf16 activation x [B, H, W, C]
 * (C % 8 == 0): cols [B * Ho * Wo, 9 * C] with cols[(b, ho, wo), (kh, kw, c)] = x[b, s ho + kh - 1, s wo + kw - 1, c], zero
 * outside the image; Ho = (H - 1) / s + 1.  With it the small-spatial 3x3 convolutions of the encoder (layer3 / layer4 of
 * the torchvision bodies, dmm/modules/vision.py:6-38, and the 3x3 heads, base.py:35-54) run as dmm_conv1x1_bf16 on the
 * patch matrix -- one GEMM with bias (+ residual) + ReLU in its epilogue -- where that beats MIOpen's convolution + a
 * separate bias / ReLU pass (encoder.FastEncoder times both once per shape). */
DMM_API int dmm_im2col3x3_bf16(const void *x, int B, int H, int W, int C, int stride, void *cols, dmm_stream_t stream);

/* (9b) A 1x1 convolution of the channels-last inference encoder with its whole tail as ONE hipBLASLt GEMM:
 *   y[rows, cout] = relu?( x[rows, cin] . w[cin, cout] + bias[cout] (+ residual[rows, cout]) )
 * x, w, residual, y bfloat16 row-major and dense, bias fp32, fp32 accumulation, one rounding.  Replaces conv1 / conv3 /
 * downsample of the torchvision bottlenecks (dmm/modules/vision.py:6-38) and the 1x1 head convolutions (base.py:35-54)
 * after BatchNorm folding; the residual rides as the GEMM's C operand (beta = 1) instead of a separate pass.
 * workspace: caller-owned scratch (may be NULL / 0: kernels that need one are then not considered);
 * DMM_ERR_UNSUPPORTED when the library offers no kernel for the shape (callers fall back to torch.mm + (9)). */
DMM_API int dmm_conv1x1_bf16(const void *x, const void *w, const float *bias, const void *residual, int64_t rows, int cin,
                             int cout, int relu, void *y, void *workspace, size_t workspace_bytes, dmm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * (10) Training-mode BatchNorm (+ residual) (+ ReLU) of the encoder, channels-last bf16, fp32 statistics: two launches each
 * way where the stock path issues five or six (three BatchNorm kernels + add + clamp).  Replaces, in the trainer's step
 * (train.py:296-307), bn -> relu and bn -> (+ identity) -> relu of the torchvision blocks (dmm/modules/vision.py:6-38),
 * the conv -> BN -> ReLU -> conv -> BN heads (base.py:43-54) and the skip projections' bn (model_encoder.py:137-140).
 * x, residual, y, dy, dx, dres: [rows, C] bfloat16 (C % 8 == 0 and 256 % (C / 8) == 0, else DMM_ERR_UNSUPPORTED);
 * everything else fp32.  Forward: dmm_bn_stats_bf16 accumulates sum(x), sum(x^2) per channel into stats [2, C] (the CALLER
 * zeroes it beforehand; atomics, order free); dmm_bn_apply_bf16 turns them into mean / invstd (biased variance, eps inside
 * the root), writes y = act(x * w * invstd + (b - mean * w * invstd) (+ residual)), saved [2, C] = (mean, invstd) and -- when
 * running_mean / running_var are given -- running <- (1 - momentum) * running + momentum * batch with the UNBIASED variance
 * (torch.nn.BatchNorm2d in training mode).  Backward: with g = dy * [y > 0] when relu, else dy: dmm_bn_bwd_reduce_bf16
 * accumulates sum(g), sum(g * xhat) into sums [2, C] (caller-zeroed); dmm_bn_bwd_dx_bf16 writes
 * dx = w * invstd * (g - mean(g) - xhat * mean(g * xhat)), dres = g (NULL: no residual branch), dweight = sum(g * xhat),
 * dbias = sum(g).  relu: 0 none; 1 the mask from the output y; 2 (no residual: dres NULL) the mask recomputed from x with the forward's
 * own fma, x * (w * invstd) + (b - mean * w * invstd) > 0 -- weight / bias as the forward saw them; y is then not read at all.
 * ------------------------------------------------------------------------------------------- */
DMM_API int dmm_bn_stats_bf16(const void *x, int64_t rows, int C, float *stats, dmm_stream_t stream);
DMM_API int dmm_bn_apply_bf16(const void *x, const void *residual, int64_t rows, int C, const float *stats,
                              const float *weight, const float *bias, float *running_mean, float *running_var,
                              float momentum, float eps, int relu, void *y, float *saved, dmm_stream_t stream);
DMM_API int dmm_bn_bwd_reduce_bf16(const void *dy, const void *x, const void *y, int64_t rows, int C, const float *saved,
                                   const float *weight, const float *bias, int relu, float *sums, dmm_stream_t stream);
DMM_API int dmm_bn_bwd_dx_bf16(const void *dy, const void *x, const void *y, int64_t rows, int C, const float *saved,
                               const float *weight, const float *bias, const float *sums, int relu, void *dx, void *dres,
                               float *dweight, float *dbias, dmm_stream_t stream);
/* The same four with `groups` STATISTICS GROUPS: the rows are `groups` consecutive ranges of rows / groups rows, each normalised
 * with its own batch statistics -- one launch computes what `groups` calls of the layer on the ranges, in order, compute: the
 * reference's trainer calls the encoder once per frame of a clip (trainer.py:95-131: BatchNorm statistics over the videos of ONE
 * frame step), and a clip batched into one encoder call keeps exactly those statistics with groups = frames.  stats / saved / sums
 * are [groups][2][C]; the running statistics take `groups` momentum updates in group order; dweight / dbias are the sums over the
 * groups.  rows % groups == 0, 1 <= groups <= 64.  (The plain entries are groups = 1.)  The two backward entries take a SECOND
 * gradient plane dy2 (may be NULL): the layer's output went to two consumers (a residual block's first convolution and its identity
 * branch, dmm/modules/vision.py:26-38) and the gradient is dy + dy2, added in fp32 inside the kernels instead of by a launch. */
DMM_API int dmm_bn_stats_grouped_bf16(const void *x, int64_t rows, int C, int groups, float *stats, dmm_stream_t stream);
DMM_API int dmm_bn_apply_grouped_bf16(const void *x, const void *residual, int64_t rows, int C, int groups, const float *stats,
                                      const float *weight, const float *bias, float *running_mean, float *running_var,
                                      float momentum, float eps, int relu, void *y, float *saved, dmm_stream_t stream);
DMM_API int dmm_bn_bwd_reduce_grouped_bf16(const void *dy, const void *dy2, const void *x, const void *y, int64_t rows, int C, int groups,
                                           const float *saved, const float *weight, const float *bias, int relu, float *sums,
                                           dmm_stream_t stream);
DMM_API int dmm_bn_bwd_dx_grouped_bf16(const void *dy, const void *dy2, const void *x, const void *y, int64_t rows, int C, int groups,
                                       const float *saved, const float *weight, const float *bias, const float *sums, int relu,
                                       void *dx, void *dres, float *dweight, float *dbias, dmm_stream_t stream);

/* (10b) Weight gradient of the encoder's convolutions (channels-last bf16 activations, fp32 gradient), what autograd
 * computes for conv1 / conv3 / downsample (1x1) and conv2 / the heads (3x3, padding 1) of dmm/modules/vision.py:6-38 and
 * base.py:35-54 under the trainer's backward (train.py:296-307):
 *   dmm_wgrad_bf16:     dw[co, ci]         = sum_r dy[r, co] * x[r, ci]            dy [rows, ldy], x [rows, ldx] (element strides)
 *   dmm_wgrad3x3_bf16:  dw[co, ci, kh, kw] = sum_(b,ho,wo) dy[(b,ho,wo), co] * x[b, s*ho+kh-1, s*wo+kw-1, ci]   (zero outside)
 *     dy [B*Ho*Wo, co], x [B, H, W, ci] dense, Ho = (H-1)/s + 1, s in {1, 2}; the patch matrix is never materialised; dw comes
 *     out in the parameter's own [co, ci, 3, 3] contiguous layout.
 * dw is fp32 and OVERWRITTEN (no zeroing, no atomics, deterministic: row slabs -> partial tables in the caller's workspace
 * -> summed in slab order).  workspace: dmm_wgrad_workspace_bytes(rows, co, cv) bytes (cv = ci, or 9 * ci for the 3x3 form;
 * 0 when the shape is not taken).  co % 64 == 0 and ci % 64 == 0 (every width of the ResNet bodies and 128-wide heads), else
 * DMM_ERR_UNSUPPORTED (callers fall back to the library product).  MFMA 32x32x16 bf16, fp32 accumulation; memory bound. */
/* (10c) Weight and layout helpers of the training encoder (dmm/modules/vision.py:6-38 under train.py:296-307).
 * dmm_wprep3x3_bf16: every 3x3 weight of a segment in one launch.  `table` = n device records of 40 bytes
 * {const float *src; uint16 *dst; uint16 *dstT; int32 co; int32 ci; int64 tile0}: src the fp32 master [co, ci, 3, 3], dst the bf16
 * channels-last weight [co, kh, kw, ci], dstT (may be null) [ci, 2-kh, 2-kw, co] -- the weight with which the DATA gradient of a
 * stride 1 / padding 1 convolution is itself a forward convolution, dX = conv(dY, dstT) (MIOpen's forward kernel for that problem
 * is ~1.8x faster than its backward-data kernel on gfx950).  co, ci multiples of 32, dst / dstT 16-byte aligned; tile0 = sum of (co/32)*(ci/32) of the records
 * before; tiles = that sum over all records.
 * dmm_subsample2_bf16: y[b, ho, wo, :] = x[b, 2ho, 2wo, :] on channels-last bf16 (the stride of a 1x1 downsample convolution);
 * dmm_upsample2_zero_bf16: its gradient, dx [B, H, W, C] written once (dy at even positions, zero elsewhere).  C % 8 == 0. */
DMM_API int dmm_wprep3x3_bf16(const void *table, int n, int64_t tiles, dmm_stream_t stream);
/* fp32 -> bf16 copies of n tensors in one launch (the 1x1 weights of a segment): device records of 32 bytes
 * {const float *src; uint16 *dst; int64 n; int64 block0}, n a multiple of 8, both 16-byte aligned; a workgroup converts 8192
 * elements: block0 = sum of ceil(n / 8192) of the records before, blocks = that sum over all records. */
DMM_API int dmm_cast_many_bf16(const void *table, int n, int64_t blocks, dmm_stream_t stream);
DMM_API int dmm_subsample2_bf16(const void *x, int B, int H, int W, int C, void *y, dmm_stream_t stream);
DMM_API int dmm_upsample2_zero_bf16(const void *dy, int B, int H, int W, int C, void *dx, dmm_stream_t stream);
DMM_API size_t dmm_wgrad_workspace_bytes(int64_t rows, int co, int cv);
DMM_API int dmm_wgrad_bf16(const void *dy, const void *x, int64_t rows, int co, int ci, int64_t ldy, int64_t ldx, float *dw,
                           void *workspace, size_t workspace_bytes, dmm_stream_t stream);
DMM_API int dmm_wgrad3x3_bf16(const void *dy, const void *x, int B, int H, int W, int ci, int co, int stride, float *dw,
                              void *workspace, size_t workspace_bytes, dmm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * (11) HIP-graph hygiene for captured steps that contain other libraries' launches (MIOpen, hipBLASLt, torch): between the
 * end of a stream capture and hipGraphInstantiate, replace every memset node (flags & 1; element size 1 / 2 / 4) and every
 * 1-D device-to-device memcpy node (flags & 2) of `graph` (a hipGraph_t) by a kernel node with the same dependencies and
 * dependents.  On the runtime of this image a replayed memset node is not reliably ordered before the kernel node behind it
 * (profiles/r04_graph_memset_node.txt; round 6: MIOpen's bf16 weight-gradient solvers clear their split-K workspace that
 * way).  n_memset / n_memcpy: nodes replaced; n_left: memset / memcpy nodes left as they were (other kinds, flags off).
 * Nothing in the reference corresponds to it (train.py:296-307 launches eagerly); used by dmm_net_amd/train_encoder.py.
 * ------------------------------------------------------------------------------------------- */
DMM_API int dmm_graph_nodes_to_kernels(void *graph, int flags, int *n_memset, int *n_memcpy, int *n_left);

#ifdef __cplusplus
}
#endif
#endif /* DMM_MATCH_H */
