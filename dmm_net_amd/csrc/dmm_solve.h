// dmm_solve.h -- shared between the two mappings of the relaxed-assignment solver:
//   dmm_solve.hip     thread = column, one wave(-group) per frame   (throughput form: thousands of frames in flight)
//   dmm_solve_rs.hip  row-split, RG x CG waves per frame            (latency form / wide tables)
// Both issue the same fp32 operations in the same order (reference relax_match.py:36-105): bit-identical results.
#pragma once
#include <stdlib.h>

#include "dmm_torch_order.h"

namespace dmm {

struct RelaxParams {
    int max_iter, proj_iter;
    float lr;
};

// Which mapping serves a launch of B frames with M template rows and solver width Pp.  The row-split form was built
// to cut the latency of a single frame's dependent chain; measured, the two tie at 10 x 50 and
// row-split wins ~10 % from ~12 rows up while few frames are in flight, so that is where it is used.
// DMM_OPT_SOLVER_KERNEL = 0 forces thread-per-column, 1 row-split wherever it is compiled (Pp <= 64); the tests run every
// solver golden through both.
bool use_row_split(int B, int M, int Pp);

int launch_relax_match_rs(const float *cos_in, const int32_t *inter, const int32_t *area_p, const int32_t *area_t,
                          const float *score_p, int B, int N, int M, const int32_t *n_valid, const int32_t *m_valid,
                          float w_feat, float w_iou, RelaxParams prm, int is_test, float *sim_out, float *R_out,
                          float *Rb_out, float *match_score, float *det_score, int32_t *iters_out, float *X_final,
                          hipStream_t stream);
int launch_relax_solve_rs(const float *C, int B, int n, int m, const int32_t *rows_valid, const int32_t *cols_valid,
                          RelaxParams prm, float *X_final, float *R_out, float *cost_out, int32_t *iters_out,
                          hipStream_t stream);

// dmm_relax_match_f32 with the table-clearing request of dmm_match_forward_ws (dmm_solve.hip)
int relax_match_launch(const float *cos_in, const int32_t *inter, const int32_t *area_p, const int32_t *area_t,
                       const float *score_p, int B, int N, int M, const int32_t *n_valid, const int32_t *m_valid,
                       float score_weight, int max_iter, int proj_iter, float lr, int is_test, float *sim_out, float *R_out,
                       float *Rb_out, float *match_score, float *det_score, int32_t *iters_out, float *X_final,
                       int clear_tables, int *cleared, dmm_stream_t stream, void *tape = nullptr, int *taped = nullptr);
// the training call's solver tape (dmm_solve.hip): bytes (0 = not taped), and the backward that walks it
size_t relax_tape_bytes(int B, int N, int M, int max_iter, int proj_iter);
int relax_match_bwd_launch(const float *sim, const float *score_p, int B, int N, int M, const int32_t *n_valid,
                           const int32_t *m_valid, int max_iter, int proj_iter, float lr, int is_test, const float *dRb,
                           const float *d_match_score, const float *d_det_score, float *dsim_out, void *workspace,
                           size_t workspace_bytes, const void *fwd_tape, const float *R_saved, const int32_t *iters_saved,
                           dmm_stream_t stream);

// General forms for tables outside the compiled envelope (dmm_wide.hip): any N, M; the solver keeps its state in
// `scratch` (B x wide_scratch_floats(M, max(N, M + 1)) floats).
size_t wide_scratch_floats(int M, int PpS);
int launch_cosine_wide(const float *featn_t, const float *featn_p, int B, int N, int M, int D, const int32_t *n_valid,
                       const int32_t *m_valid, float *cos_out, hipStream_t stream);
int launch_relax_match_wide(const float *cos_in, const int32_t *inter, const int32_t *area_p, const int32_t *area_t,
                            const float *score_p, int B, int N, int M, const int32_t *n_valid, const int32_t *m_valid,
                            float w_feat, float w_iou, RelaxParams prm, int is_test, float *sim_out, float *R_out,
                            float *Rb_out, float *match_score, float *det_score, int32_t *iters_out, float *X_final,
                            float *scratch, hipStream_t stream);

// backward of the solver at any table size (dmm_wide.hip): workspace bytes PER FRAME, and the launch
size_t wide_bwd_bytes(int N, int M, int max_iter, int proj_iter);
int launch_relax_match_bwd_wide(const float *sim, const float *score_p, int B, int N, int M, const int32_t *n_valid,
                                const int32_t *m_valid, RelaxParams prm, int is_test, const float *dRb, const float *dms,
                                const float *dds, float *dsim_out, void *workspace, hipStream_t stream);

}  // namespace dmm

namespace dmm {
int opt(int key);
// Threads per workgroup: the one-wave solver gets a second wave (the cost-norm helper, norm_helper_wave) while few frames
// are in flight -- two waves per frame then still sit on different SIMDs; DMM_OPT_SOLVER_HELPER_MAX (default 512 frames,
// 0 = never) moves the switch.
static inline int solver_block(int ng, int B) {
    if (ng != 1) return 64 * ng;
    const int helper_max = opt(DMM_OPT_SOLVER_HELPER_MAX);
    return B <= helper_max ? 128 : 64;
}
}  // namespace dmm

// Kernel selection: exact-row-count instantiations for the common small problems (one wave per
// frame), guarded generic ones (MT in {8,16,32}) otherwise.
#define DMM_DISPATCH_SOLVER(M_, W_, EXACT_OK, CALL)                                                          \
    do {                                                                                                     \
        const int ng_ = ((W_) + 63) / 64;                                                                    \
        if (ng_ <= 1 && (EXACT_OK)) {                                                                        \
            switch (M_) {                                                                                    \
                case 1: CALL(1, 1, true); break;   case 2: CALL(2, 1, true); break;                          \
                case 3: CALL(3, 1, true); break;   case 4: CALL(4, 1, true); break;                          \
                case 5: CALL(5, 1, true); break;   case 6: CALL(6, 1, true); break;                          \
                case 7: CALL(7, 1, true); break;   case 8: CALL(8, 1, true); break;                          \
                case 9: CALL(9, 1, true); break;   case 10: CALL(10, 1, true); break;                        \
                case 11: CALL(11, 1, true); break; case 12: CALL(12, 1, true); break;                        \
                case 13: CALL(13, 1, true); break; case 14: CALL(14, 1, true); break;                        \
                case 15: CALL(15, 1, true); break; case 16: CALL(16, 1, true); break;                        \
                default: CALL(32, 1, false); break;                                                          \
            }                                                                                                \
        } else if (ng_ <= 1) {                                                                               \
            if ((M_) <= 8) CALL(8, 1, false); else if ((M_) <= 16) CALL(16, 1, false); else CALL(32, 1, false); \
        } else if (ng_ == 2) {                                                                               \
            if ((M_) <= 8) CALL(8, 2, false); else if ((M_) <= 16) CALL(16, 2, false); else CALL(32, 2, false); \
        } else {                                                                                             \
            if ((M_) <= 8) CALL(8, 4, false); else if ((M_) <= 16) CALL(16, 4, false);                       \
            else if ((M_) == 20 && (EXACT_OK)) CALL(20, 4, true); else CALL(32, 4, false);                   \
        }                                                                                                    \
    } while (0)

