// dmm_solve_plain.hip -- relax_matching on a bare cost matrix (dmm_relax_solve_f32): the kernels of dmm_solve.hip without the
// similarity prologue and the score epilogue.  A translation unit of its own: the one-wave core is compiled per row count,
// per width class of the model's frames and with the reference's 5 projection sweeps as straight-line code, and one file
// with all of it set the build's critical path.
#include "dmm_solve_core.h"

namespace dmm {

// Solver-only kernel on a caller-provided C [B, n, m].
template <int MT, int NG, bool EXACT>
__global__ __launch_bounds__(NG == 1 ? 128 : 64 * NG) void relax_solve_kernel(const float *__restrict__ Cin, int n_max, int m_max,
                                                              const int32_t *__restrict__ rows_valid,
                                                              const int32_t *__restrict__ cols_valid,
                                                              RelaxParams prm, float *__restrict__ X_final,
                                                              float *__restrict__ R_out, float *__restrict__ cost_out,
                                                              int32_t *__restrict__ iters_out) {
    __shared__ float red_buf[2 * NG * (MT + 1)];
    __shared__ float xbuf[MT * 64 * NG];
    __shared__ float rsbuf[MT + 1];
    __shared__ int hs[4];
    if (NG == 1 && solver_helper_entry(xbuf, hs)) return;
    const int b = blockIdx.x, col = threadIdx.x;
    BlockRed<MT, NG> red(red_buf, threadIdx.x >> 6);
    const int n = EXACT ? MT : (rows_valid ? rows_valid[b] : n_max);
    const int m = cols_valid ? cols_valid[b] : m_max;
    float C[MT], X[MT], acc[MT];
    if (n <= 0 || m <= 0) {                                     // dead frame
        for (int i = threadIdx.x; i < n_max * m_max; i += 64 * NG) {
            if (X_final) X_final[(int64_t)b * n_max * m_max + i] = 0.0f;
            if (R_out) R_out[(int64_t)b * n_max * m_max + i] = 0.0f;
        }
        if (iters_out && threadIdx.x == 0) iters_out[b] = 0;
        if (NG == 1) solver_helper_stop(hs);
        return;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) C[i] = (i < n && col < m) ? Cin[((int64_t)b * n_max + i) * m_max + col] : 0.0f;
    const int iters = relax_core<MT, NG, EXACT>(C, n, m, col, prm, red, xbuf, rsbuf, X, acc,
                                                cost_out ? cost_out + (int64_t)b * (prm.max_iter + 1) : nullptr,
                                                RelaxTape{nullptr, nullptr}, hs);
    const float flen = (float)(iters + 1);
    for (int i = 0; i < n_max; ++i) {
        if (col < m_max) {
            const bool lv = i < n && col < m;
            float xv = 0.0f, rv = 0.0f;
#pragma unroll
            for (int k = 0; k < MT; ++k)
                if (k == i) { xv = X[k]; rv = acc[k] / flen; }
            if (X_final) X_final[((int64_t)b * n_max + i) * m_max + col] = lv ? xv : 0.0f;
            if (R_out) R_out[((int64_t)b * n_max + i) * m_max + col] = lv ? rv : 0.0f;
        }
    }
    if (iters_out && threadIdx.x == 0) iters_out[b] = iters;
}


}  // namespace dmm

extern "C" int dmm_relax_solve_f32(const float *C, int B, int n, int m, const int32_t *rows_valid,
                                   const int32_t *cols_valid, int max_iter, int proj_iter, float lr,
                                   float *X_final, float *R_out, float *cost_out, int32_t *iters_out,
                                   dmm_stream_t stream) {
    if (B < 0 || n <= 0 || m <= 0 || max_iter < 0 || proj_iter < 0) return DMM_ERR_BAD_ARG;
    if (B == 0) return DMM_OK;
    if (!C) return DMM_ERR_BAD_ARG;
    if (n > DMM_MAX_TEMPLATES || m > DMM_MAX_PROPOSALS) return DMM_ERR_UNSUPPORTED;
    const dmm::RelaxParams prm{max_iter, proj_iter, lr};
    if (dmm::use_row_split(B, n, m))
        return dmm::launch_relax_solve_rs(C, B, n, m, rows_valid, cols_valid, prm, X_final, R_out, cost_out, iters_out,
                                          (hipStream_t)stream);
#define DMM_CALL(MT_, NG_, EX_)                                                                                     \
    hipLaunchKernelGGL((dmm::relax_solve_kernel<MT_, NG_, EX_>), dim3(B), dim3(dmm::solver_block(NG_, B)), 0, (hipStream_t)stream, C, \
                       n, m, rows_valid, cols_valid, prm, X_final, R_out, cost_out, iters_out)
    DMM_DISPATCH_SOLVER(n, m, rows_valid == nullptr, DMM_CALL);
#undef DMM_CALL
    return dmm::check_launch();
}

