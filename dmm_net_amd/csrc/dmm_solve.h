// dmm_solve.h -- shared between the two mappings of the relaxed-assignment solver:
//   dmm_solve.hip     thread = column, one wave(-group) per frame   (throughput form: thousands of frames in flight)
//   dmm_solve_rs.hip  row-split, RG x CG waves per frame            (latency form / wide tables)
// Both issue the same fp32 operations in the same order (reference relax_match.py:36-105): bit-identical results.
#pragma once
#include "dmm_torch_order.h"

namespace dmm {

struct RelaxParams {
    int max_iter, proj_iter;
    float lr;
};

// Which mapping serves a launch of B frames with M template rows and solver width Pp:
// the row-split form when few frames are in flight (a frame's dependent chain is the latency) or when the table is
// wide (thread-per-column then needs 256 VGPRs per wave and cannot share a SIMD with the streaming kernels).
// DMM_SOLVER_KERNEL=0 forces thread-per-column, =1 row-split (tests run every golden through both).
bool use_row_split(int B, int M, int Pp);

int launch_relax_match_rs(const float *cos_in, const int32_t *inter, const int32_t *area_p, const int32_t *area_t,
                          const float *score_p, int B, int N, int M, const int32_t *n_valid, const int32_t *m_valid,
                          float w_feat, float w_iou, RelaxParams prm, int is_test, float *sim_out, float *R_out,
                          float *Rb_out, float *match_score, float *det_score, int32_t *iters_out, float *X_final,
                          hipStream_t stream);
int launch_relax_solve_rs(const float *C, int B, int n, int m, const int32_t *rows_valid, const int32_t *cols_valid,
                          RelaxParams prm, float *X_final, float *R_out, float *cost_out, int32_t *iters_out,
                          hipStream_t stream);

}  // namespace dmm
