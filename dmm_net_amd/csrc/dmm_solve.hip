// dmm_solve.hip -- similarity + relaxed assignment (projected gradient + Dykstra) on gfx950.
//
// Replaces, per frame, the reference's
//   get_cosine_score               dmm/utils/match_helper.py:51-64
//   sim mix + padding              dmm/modules/match_model.py:89-90, :107-116
//   relax_matching / project_row / project_col
//                                  dmm/modules/submodules/relax_match.py:9-105
//   mean(X_list), logic mask, scores   dmm/modules/match_model.py:118-130, :146-147
// which in the reference are ~17 eager kernels + >= 1 host sync (.item()) per inner sweep.
//
// Solver mapping: one workgroup of NG waves per frame; thread j owns COLUMN j (a proposal) of the
// [M, Pp] problem and keeps its M-row slice of C, X, the three Dykstra increments P0..P2 and the
// running sum of iterates in REGISTERS for all max_iter x proj_iter sweeps -- nothing touches HBM
// between the prologue and the epilogue.  Elementwise fp32 ops are issued exactly as the reference's
// eager ops (compiled with -ffp-contract=off); divisions by the constant row/column counts use an
// exactly-rounded reciprocal refinement (div_by_const); column sums (in-lane), row sums and the cost
// norm (through LDS, aligned 8-lane groups playing the AVX2 lanes) follow the SUMMATION ORDER of the
// torch CPU kernels the reference was captured with (dmm_torch_order.h), so every iterate, both
// data-dependent early exits (relax_match.py:88-89, :96-98) and the executed-iteration count are bit
// exact against the reference.
//
// Roofline: neither HBM nor MFMA.  One wave per frame issues ~145 (5 rows) to ~230 (10 rows) instructions per projection
// sweep at ~5-6 cycles each (one wave on a SIMD: nothing hides a dependent VALU op's 6 cycles, a DPP add's 13, an LDS
// turn-around's 76 -- tools/clock_probe.py), so the sweep is ISSUE bound and every compile-time fact the scheduler gets
// pays: the row count (exact instantiations), the width class of the model's frames, the reference's 5 sweeps as
// straight-line code (dmm_solve_core.h).  Throughput comes from running one frame per wave on all 1024 SIMDs.
#include "dmm_solve_core.h"

namespace dmm {

bool use_row_split(int B, int M, int Pp) {
    const int e = opt(DMM_OPT_SOLVER_KERNEL);  // -1 by shape; the solver goldens pin 0 and 1
    if (Pp > 64) return false;                  // the row-split form is compiled for one column group (Pp <= 64)
    if (e == 0) return false;
    if (e == 1) return true;
    // Round 3: the one-wave latency form (relax_core_w1 + the cost-norm helper wave) overtook it everywhere it is compiled
    // (tools/solver_timing.py, us per solve at 20 x 5, one-wave vs row-split: 5 x 50: 51 vs 82, 10 x 50: 70 vs 99,
    // 16 x 64: 93 vs 112); it stays as the alternative mapping every solver golden is also run through.
    return false;
}


template <int MT, int NG, bool EXACT>
// (4-wave instantiations up to 20 rows are capped at 256 registers -- 4 spilled -- so that two frames share a CU:
// 20 x 200 needed 256 VGPRs + 12 AGPRs = one wave per SIMD, i.e. 256 frames filled the chip and 512 took twice as long)
__global__ __launch_bounds__(NG == 1 ? 128 : 64 * NG, (NG == 4 && MT <= 20) ? 2 : 1) void relax_match_kernel(
    const float *__restrict__ cos_in, const int32_t *__restrict__ inter, const int32_t *__restrict__ area_p,
    const int32_t *__restrict__ area_t, const float *__restrict__ score_p, int N, int M,
    const int32_t *__restrict__ n_valid, const int32_t *__restrict__ m_valid, float w_feat, float w_iou,
    RelaxParams prm, int is_test, float *__restrict__ sim_out, float *__restrict__ R_out, float *__restrict__ Rb_out,
    float *__restrict__ match_score, float *__restrict__ det_score, int32_t *__restrict__ iters_out,
    float *__restrict__ X_final) {
    __shared__ float red_buf[2 * NG * (MT + 1)];
    __shared__ float xbuf[MT * 64 * NG];
    __shared__ float rsbuf[MT + 1];
    __shared__ int hs[4];
    if (NG == 1 && solver_helper_entry(xbuf, hs)) return;       // 128-thread workgroups: wave 1 is the norm helper
    relax_match_body<MT, NG, EXACT>(cos_in, inter, area_p, area_t, score_p, N, M, n_valid, m_valid, w_feat, w_iou, prm,
                                    is_test, sim_out, R_out, Rb_out, match_score, det_score, iters_out, X_final, red_buf,
                                    xbuf, rsbuf, hs);
    if (NG == 1) solver_helper_stop(hs);                        // (paths that never reached the solver)
}

// The same kernel for the TRAINING forward (dmm_match_train_forward): one wave per frame, and the solver records its sweeps
// (gate bits per column and sweep, executed sweeps per outer iteration) for the backward.  Kernels of their own so that the
// evaluator's solve stays the code it was.
template <int MT, bool EXACT>
__global__ __launch_bounds__(128) void relax_match_taped_kernel(
    const float *__restrict__ cos_in, const int32_t *__restrict__ inter, const int32_t *__restrict__ area_p,
    const int32_t *__restrict__ area_t, const float *__restrict__ score_p, int N, int M,
    const int32_t *__restrict__ n_valid, const int32_t *__restrict__ m_valid, float w_feat, float w_iou,
    RelaxParams prm, int is_test, float *__restrict__ sim_out, float *__restrict__ R_out, float *__restrict__ Rb_out,
    float *__restrict__ match_score, float *__restrict__ det_score, int32_t *__restrict__ iters_out,
    uint2 *__restrict__ tape_bits, int *__restrict__ tape_sweeps) {
    __shared__ float red_buf[2 * (MT + 1)];
    __shared__ float xbuf[MT * 64];
    __shared__ float rsbuf[MT + 1];
    __shared__ int hs[4];
    if (solver_helper_entry(xbuf, hs)) return;
    relax_match_body<MT, 1, EXACT, false, true>(cos_in, inter, area_p, area_t, score_p, N, M, n_valid, m_valid, w_feat, w_iou,
                                                prm, is_test, sim_out, R_out, Rb_out, match_score, det_score, iters_out,
                                                nullptr, red_buf, xbuf, rsbuf, hs, tape_bits, tape_sweeps);
    solver_helper_stop(hs);
}

// Ragged batches of small problems (the product: up to maxseqlen = 5 templates per video, a different count per video):
// one wave per frame picks the EXACT-row-count body of ITS frame.  The guarded MT = 8 instantiation carried 8 rows and a
// row guard on every element for every frame (5 templates, eval setting 40 x 5: 183 us per solve; exact: ~120).
template <int MTMAX>
__global__ __launch_bounds__(128) void relax_match_ragged_kernel(
    const float *__restrict__ cos_in, const int32_t *__restrict__ inter, const int32_t *__restrict__ area_p,
    const int32_t *__restrict__ area_t, const float *__restrict__ score_p, int N, int M,
    const int32_t *__restrict__ n_valid, const int32_t *__restrict__ m_valid, float w_feat, float w_iou,
    RelaxParams prm, int is_test, float *__restrict__ sim_out, float *__restrict__ R_out, float *__restrict__ Rb_out,
    float *__restrict__ match_score, float *__restrict__ det_score, int32_t *__restrict__ iters_out,
    float *__restrict__ X_final) {
    __shared__ float red_buf[2 * (MTMAX + 1)];
    __shared__ float xbuf[MTMAX * 64];
    __shared__ float rsbuf[MTMAX + 1];
    __shared__ int hs[4];
    if (solver_helper_entry(xbuf, hs)) return;
    const int Mb = m_valid ? m_valid[blockIdx.x] : M;
#define DMM_BODY(K)                                                                                                     \
    case K:                                                                                                             \
        if constexpr (K <= MTMAX)                                                                                       \
            relax_match_body<K, 1, true>(cos_in, inter, area_p, area_t, score_p, N, M, n_valid, m_valid, w_feat, w_iou, \
                                         prm, is_test, sim_out, R_out, Rb_out, match_score, det_score, iters_out,       \
                                         X_final, red_buf, xbuf, rsbuf, hs);                                            \
        break;
    switch (Mb) {
        DMM_BODY(2) DMM_BODY(3) DMM_BODY(4) DMM_BODY(5) DMM_BODY(6) DMM_BODY(7) DMM_BODY(8)
        default:                                            // 1 template, and dead frames (Mb <= 0: zeros)
            relax_match_body<1, 1, false>(cos_in, inter, area_p, area_t, score_p, N, M, n_valid, m_valid, w_feat, w_iou, prm,
                                          is_test, sim_out, R_out, Rb_out, match_score, det_score, iters_out, X_final,
                                          red_buf, xbuf, rsbuf, hs);
            break;
    }
#undef DMM_BODY
    solver_helper_stop(hs);
}

// ... and its taped twin for the training forward (see relax_match_taped_kernel)
template <int MTMAX>
__global__ __launch_bounds__(128) void relax_match_ragged_taped_kernel(
    const float *__restrict__ cos_in, const int32_t *__restrict__ inter, const int32_t *__restrict__ area_p,
    const int32_t *__restrict__ area_t, const float *__restrict__ score_p, int N, int M,
    const int32_t *__restrict__ n_valid, const int32_t *__restrict__ m_valid, float w_feat, float w_iou,
    RelaxParams prm, int is_test, float *__restrict__ sim_out, float *__restrict__ R_out, float *__restrict__ Rb_out,
    float *__restrict__ match_score, float *__restrict__ det_score, int32_t *__restrict__ iters_out,
    uint2 *__restrict__ tape_bits, int *__restrict__ tape_sweeps) {
    __shared__ float red_buf[2 * (MTMAX + 1)];
    __shared__ float xbuf[MTMAX * 64];
    __shared__ float rsbuf[MTMAX + 1];
    __shared__ int hs[4];
    if (solver_helper_entry(xbuf, hs)) return;
    const int Mb = m_valid ? m_valid[blockIdx.x] : M;
#define DMM_BODY(K)                                                                                                     \
    case K:                                                                                                             \
        if constexpr (K <= MTMAX)                                                                                       \
            relax_match_body<K, 1, true, false, true>(cos_in, inter, area_p, area_t, score_p, N, M, n_valid, m_valid, w_feat, \
                                                      w_iou, prm, is_test, sim_out, R_out, Rb_out, match_score, det_score,   \
                                                      iters_out, nullptr, red_buf, xbuf, rsbuf, hs, tape_bits, tape_sweeps); \
        break;
    switch (Mb) {
        DMM_BODY(2) DMM_BODY(3) DMM_BODY(4) DMM_BODY(5) DMM_BODY(6) DMM_BODY(7) DMM_BODY(8)
        default:
            relax_match_body<1, 1, false, false, true>(cos_in, inter, area_p, area_t, score_p, N, M, n_valid, m_valid, w_feat,
                                                       w_iou, prm, is_test, sim_out, R_out, Rb_out, match_score, det_score,
                                                       iters_out, nullptr, red_buf, xbuf, rsbuf, hs, tape_bits, tape_sweeps);
            break;
    }
#undef DMM_BODY
    solver_helper_stop(hs);
}

// ---------------------------------------------------------------------------------------------
// Backward of the layer kernel with respect to sim (reference: torch autograd through relax_matching,
// relax_match.py:68-98, and match_model.py:121-147).  One workgroup per frame:
//   1. re-run the forward solver from the saved sim (bit-identical: same code), taping per sweep the relu
//      pass bits and the column-over flag, per outer iteration the executed sweep count;
//   2. epilogue adjoints: Rb = R*logic, match_score = max clamp(R,0,1)*sim_pad, det_score = sum score*Rb
//      -> dR and the direct d sim term;
//   3. walk the tape backwards: every sweep is a linear map given its bits, the gradient step contributes
//      dC -= lr * g; the greedy init, the masks and the exits carry no gradient;
//   4. dsim = -dC on the live [M, N] block.
// ---------------------------------------------------------------------------------------------
constexpr int kMaxTapeOuter = 1024;

template <int MT, int NG, bool EXACT>
__device__ __forceinline__ void relax_match_bwd_body(
    const float *__restrict__ sim_in, const float *__restrict__ score_p, int N, int M,
    const int32_t *__restrict__ n_valid, const int32_t *__restrict__ m_valid, RelaxParams prm, int is_test,
    const float *__restrict__ dRb_in, const float *__restrict__ dms_in, const float *__restrict__ dds_in,
    float *__restrict__ dsim_out, uint2 *__restrict__ tape_ws, float *red_buf, float *xbuf, float *rsbuf, int *sweeps_s,
    const float *__restrict__ R_saved, const int32_t *__restrict__ iters_saved) {
    // R_saved != null (one-wave kernels; dmm_match_train_backward): the forward kept its tape -- tape_ws holds the sweep
    // records AND, behind them, the sweep counts ([B][max_iter]); R and the iteration counts are the forward's outputs.
    // Step 1 (the re-run) is skipped.
    const int b = blockIdx.x;
    const int col = threadIdx.x;
    BlockRed<MT, NG> red(red_buf, threadIdx.x >> 6);
    const int Nb = n_valid ? n_valid[b] : N;
    const int Mb = EXACT ? MT : (m_valid ? m_valid[b] : M);
    const int PpS = N > M ? N : M + 1;
    float *dsim_b = dsim_out + (int64_t)b * M * N;
    if (Mb <= 0 || Nb <= 0) {
        for (int i = threadIdx.x; i < M * N; i += 64 * NG) dsim_b[i] = 0.0f;
        return;
    }
#define DMM_ROW(i) (EXACT || (i) < Mb)
    const int Pp = Nb > Mb ? Nb : Mb + 1;
    const bool has_prop = col < Nb;
    const bool livec = col < Pp;
    float C[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const float sv = (DMM_ROW(i) && has_prop) ? sim_in[(int64_t)b * M * N + (int64_t)i * N + col] : 0.0f;
        C[i] = (DMM_ROW(i) && livec) ? -sv : 0.0f;
    }
    float X[MT], acc[MT];
    RelaxTape tape{tape_ws + (size_t)b * prm.max_iter * prm.proj_iter * (64 * NG), sweeps_s};
    int iters;
    const bool from_tape = NG == 1 && R_saved != nullptr;
    if (from_tape) {
        iters = iters_saved[b];
        const int *sweeps_g = (const int *)(tape_ws + (size_t)gridDim.x * prm.max_iter * prm.proj_iter * 64) + (size_t)b * prm.max_iter;
        for (int i = threadIdx.x; i < iters; i += 64 * NG) sweeps_s[i] = sweeps_g[i];
        const int PpR = N > M ? N : M + 1;
#pragma unroll
        for (int i = 0; i < MT; ++i)
            acc[i] = (DMM_ROW(i) && livec) ? R_saved[(int64_t)b * M * PpR + (int64_t)i * PpR + col] : 0.0f;
    } else {
        iters = relax_core<MT, NG, EXACT, true>(C, Mb, Pp, col, prm, red, xbuf, rsbuf, X, acc, nullptr, tape);
    }
    __syncthreads();                                                   // sweeps_s + tape visible to the block

    // ---- epilogue adjoints -> dR (per X_list entry: g = dR / len) and the direct d(sim_pad) term ----
    const float flen = (float)(iters + 1);
    const float sc = has_prop ? score_p[(int64_t)b * N + col] : 0.0f;
    float r[MT], rmax[MT], v[MT], vmax[MT], gdirect[MT], gl[MT];
    int cand[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        r[i] = from_tape ? acc[i] : acc[i] / flen;                     // (the forward's R is this very quotient)
        rmax[i] = (livec && DMM_ROW(i)) ? r[i] : -__builtin_inff();
        const float rc = r[i] < 0.0f ? 0.0f : (r[i] > 1.0f ? 1.0f : r[i]);
        v[i] = (livec && DMM_ROW(i)) ? rc * (-C[i]) : -__builtin_inff();
        vmax[i] = v[i];
    }
    wave_max_rows<MT>(rmax);
    red.fold(rmax, fmax_op());
    wave_max_rows<MT>(vmax);
    red.fold(vmax, fmax_op());
#pragma unroll
    for (int i = 0; i < MT; ++i) cand[i] = (livec && DMM_ROW(i) && v[i] == vmax[i]) ? col : 0x7fffffff;
    wave_min_rows_i32<MT>(cand);
    red.min_i32(cand);                                                 // torch.max(dim) backward: first maximal index
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        gl[i] = 0.0f;
        gdirect[i] = 0.0f;
        if (DMM_ROW(i) && livec) {
            const float lg = is_test ? (r[i] == rmax[i] ? 1.0f : 0.0f) : (r[i] > 0.01f ? 1.0f : 0.0f);
            const float dms = dms_in ? dms_in[(int64_t)b * M + i] : 0.0f;
            const float dds = dds_in ? dds_in[(int64_t)b * M + i] : 0.0f;
            float dR = ((dRb_in ? dRb_in[(int64_t)b * M * PpS + (int64_t)i * PpS + col] : 0.0f) + dds * sc) * lg;
            if (col == cand[i]) {
                const float rc = r[i] < 0.0f ? 0.0f : (r[i] > 1.0f ? 1.0f : r[i]);
                if (r[i] >= 0.0f && r[i] <= 1.0f) dR += dms * (-C[i]);  // clamp passes its gradient on [0, 1]
                gdirect[i] = dms * rc;                                 // d/d(sim_pad) of clamp(R) * sim_pad
            }
            gl[i] = dR / flen;                                          // R = sum(X_list) / len
        }
    }

    // ---- reverse sweep through the taped iterations ----
    float gX[MT], gP0[MT], gP1[MT], gP2[MT], gC[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) { gX[i] = 0.0f; gP0[i] = 0.0f; gP1[i] = 0.0f; gP2[i] = 0.0f; gC[i] = 0.0f; }
    const float inv_m = 1.0f / (float)Pp, inv_n = 1.0f / (float)Mb;
    int pos = 0;
    for (int it = 0; it < iters; ++it) pos += sweeps_s[it];
    for (int it = iters - 1; it >= 0; --it) {
        const int ns = sweeps_s[it];
        for (int sidx = 0; sidx < ns; ++sidx) {
            --pos;
            const uint2 bits = tape.bits[(size_t)pos * (64 * NG) + threadIdx.x];
            // row projection: y2 = c - (rowsum(c) - 1)/m ; P2' = c - y2 ; X' = y2
            float gy2[MT], rs[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                gy2[i] = (DMM_ROW(i) && livec) ? gX[i] - gP2[i] : 0.0f;
                rs[i] = gy2[i];
            }
            wave_sum_rows<MT>(rs);
            red.sum(rs);
            float cg = 0.0f;
            float gy1[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                gy1[i] = 0.0f;
                if (DMM_ROW(i) && livec) {
                    const float gc = gP2[i] + gy2[i] - rs[i] * inv_m;
                    gP2[i] = gc;                                       // c = y1 + P2
                    gy1[i] = gc - gP1[i];                              // P1' = b - y1
                    cg += gy1[i];
                }
            }
            // column projection: y1 = b - over * (colsum(b) - 1)/n
            const float corr = bits.y ? cg * inv_n : 0.0f;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if (DMM_ROW(i) && livec) {
                    const float gb = gP1[i] + gy1[i] - corr;
                    gP1[i] = gb;                                       // b = y0 + P1
                    const float gy0 = gb - gP0[i];                     // P0' = a - y0
                    const float ga = gP0[i] + (((bits.x >> i) & 1u) ? gy0 : 0.0f);   // y0 = relu(a)
                    gX[i] = ga;                                        // a = X + P0
                    gP0[i] = ga;
                }
            }
        }
        // gradient step: X_pre = X_prev - lr*C, and X_pre is the X_list entry of this iteration
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if (DMM_ROW(i) && livec) {
                const float g = gX[i] + gl[i];
                gC[i] = gC[i] - prm.lr * g;
                gX[i] = g;
            }
        }
    }
    // C = -sim_pad  ->  dsim = -dC (+ the direct match_score term); padded columns are dropped
#pragma unroll
    for (int i = 0; i < MT; ++i)
        if (DMM_ROW(i) && has_prop) dsim_b[(int64_t)i * N + col] = gdirect[i] - gC[i];
#undef DMM_ROW
    for (int i = Mb; i < M; ++i)
        if (col < N) dsim_b[(int64_t)i * N + col] = 0.0f;
    if (!has_prop && col < N)
        for (int i = 0; i < Mb; ++i) dsim_b[(int64_t)i * N + col] = 0.0f;
}

template <int MT, int NG, bool EXACT>
__global__ __launch_bounds__(64 * NG) void relax_match_bwd_kernel(
    const float *__restrict__ sim_in, const float *__restrict__ score_p, int N, int M,
    const int32_t *__restrict__ n_valid, const int32_t *__restrict__ m_valid, RelaxParams prm, int is_test,
    const float *__restrict__ dRb_in, const float *__restrict__ dms_in, const float *__restrict__ dds_in,
    float *__restrict__ dsim_out, uint2 *__restrict__ tape_ws, const float *__restrict__ R_saved,
    const int32_t *__restrict__ iters_saved) {
    __shared__ float red_buf[2 * NG * (MT + 1)];
    __shared__ float xbuf[MT * 64 * NG];
    __shared__ float rsbuf[MT + 1];
    __shared__ int sweeps_s[kMaxTapeOuter];
    relax_match_bwd_body<MT, NG, EXACT>(sim_in, score_p, N, M, n_valid, m_valid, prm, is_test, dRb_in, dms_in, dds_in, dsim_out,
                                        tape_ws, red_buf, xbuf, rsbuf, sweeps_s, R_saved, iters_saved);
}

// Ragged template counts (DMM_Model's batches carry m_valid; usually every video has all of its templates): like the
// forward's relax_match_ragged_kernel, the one wave of a frame runs the EXACT-row-count body of ITS frame.  The guarded
// MT = 8 instantiation carried 8 rows and a row guard on every element of every sweep, forward re-run and reverse walk:
// 284 us for the backward of 4 videos x 5 templates at 10 x 5 against 57 us for one exact 5-row frame (round 5,
// tools/dropin_trace.py model).
template <int MTMAX>
__global__ __launch_bounds__(64) void relax_match_bwd_ragged_kernel(
    const float *__restrict__ sim_in, const float *__restrict__ score_p, int N, int M,
    const int32_t *__restrict__ n_valid, const int32_t *__restrict__ m_valid, RelaxParams prm, int is_test,
    const float *__restrict__ dRb_in, const float *__restrict__ dms_in, const float *__restrict__ dds_in,
    float *__restrict__ dsim_out, uint2 *__restrict__ tape_ws, const float *__restrict__ R_saved,
    const int32_t *__restrict__ iters_saved) {
    __shared__ float red_buf[2 * (MTMAX + 1)];
    __shared__ float xbuf[MTMAX * 64];
    __shared__ float rsbuf[MTMAX + 1];
    __shared__ int sweeps_s[kMaxTapeOuter];
    const int Mb = m_valid ? m_valid[blockIdx.x] : M;
#define DMM_BODY(K)                                                                                                       \
    case K:                                                                                                               \
        if constexpr (K <= MTMAX)                                                                                         \
            relax_match_bwd_body<K, 1, true>(sim_in, score_p, N, M, n_valid, m_valid, prm, is_test, dRb_in, dms_in, dds_in, \
                                             dsim_out, tape_ws, red_buf, xbuf, rsbuf, sweeps_s, R_saved, iters_saved);    \
        break;
    switch (Mb) {
        DMM_BODY(2) DMM_BODY(3) DMM_BODY(4) DMM_BODY(5) DMM_BODY(6) DMM_BODY(7) DMM_BODY(8)
        default:                                            // 1 template, and dead frames (Mb <= 0: zeros)
            relax_match_bwd_body<1, 1, false>(sim_in, score_p, N, M, n_valid, m_valid, prm, is_test, dRb_in, dms_in, dds_in,
                                              dsim_out, tape_ws, red_buf, xbuf, rsbuf, sweeps_s, R_saved, iters_saved);
            break;
    }
#undef DMM_BODY
}

}  // namespace dmm

extern "C" int dmm_relax_match_f32(const float *cos_in, const int32_t *inter, const int32_t *area_p,
                                   const int32_t *area_t, const float *score_p, int B, int N, int M,
                                   const int32_t *n_valid, const int32_t *m_valid, float score_weight, int max_iter,
                                   int proj_iter, float lr, int is_test, float *sim_out, float *R_out, float *Rb_out,
                                   float *match_score, float *det_score, int32_t *iters_out, float *X_final,
                                   dmm_stream_t stream) {
    return dmm::relax_match_launch(cos_in, inter, area_p, area_t, score_p, B, N, M, n_valid, m_valid, score_weight, max_iter,
                                   proj_iter, lr, is_test, sim_out, R_out, Rb_out, match_score, det_score, iters_out,
                                   X_final, 0, nullptr, stream);
}

// Bytes of the tape dmm_match_train_forward keeps for dmm_match_train_backward: per frame max_iter * proj_iter sweep records of
// 64 x 8 bytes + max_iter sweep counts; 0 = this table is not taped (wider than one wave: the backward re-runs the solver).
size_t dmm::relax_tape_bytes(int B, int N, int M, int max_iter, int proj_iter) {
    if (B <= 0 || N <= 0 || M <= 0 || max_iter <= 0 || proj_iter <= 0) return 0;
    const int Pp = N > M ? N : M + 1;
    if (M > DMM_MAX_TEMPLATES || Pp > 64 || max_iter > dmm::kMaxTapeOuter || dmm::opt(DMM_OPT_FORCE_WIDE) == 1) return 0;
    return (size_t)B * ((size_t)max_iter * proj_iter * 64 * sizeof(uint2) + (size_t)max_iter * sizeof(int));
}

// dmm_relax_match_f32 proper.  clear_tables (dmm_match_forward_ws only; the tables are then its workspace, not the
// caller's): ask the kernel to zero inter / area_p / area_t once it has read them; *cleared says whether the kernel that
// was launched does that (the thread-per-column kernels on dense frames do, the other mappings do not).
int dmm::relax_match_launch(const float *cos_in, const int32_t *inter, const int32_t *area_p, const int32_t *area_t,
                            const float *score_p, int B, int N, int M, const int32_t *n_valid, const int32_t *m_valid,
                            float score_weight, int max_iter, int proj_iter, float lr, int is_test, float *sim_out,
                            float *R_out, float *Rb_out, float *match_score, float *det_score, int32_t *iters_out,
                            float *X_final, int clear_tables, int *cleared, dmm_stream_t stream, void *tape, int *taped) {
    if (cleared) *cleared = 0;
    if (taped) *taped = 0;
    is_test = is_test != 0;                       // the upper bits of the kernels' argument are the library's own
    if (B < 0 || N < 0 || M < 0 || max_iter < 0 || proj_iter < 0) return DMM_ERR_BAD_ARG;
    if (B == 0 || M == 0) return DMM_OK;
    if (N == 0) return DMM_ERR_BAD_ARG;
    if (!cos_in || !inter || !area_p || !area_t || !score_p || !sim_out || !Rb_out || !match_score || !det_score)
        return DMM_ERR_BAD_ARG;
    const int Pp = N > M ? N : M + 1;
    if (M > DMM_MAX_TEMPLATES || Pp > DMM_MAX_PROPOSALS) return DMM_ERR_UNSUPPORTED;
    const dmm::RelaxParams prm{max_iter, proj_iter, lr};
    // python: sim*(1-w) + iou*w with w a python float -> both scalars rounded to fp32 once
    const float w_feat = (float)(1.0 - (double)score_weight), w_iou = score_weight;
    if (dmm::use_row_split(B, M, Pp))
        return dmm::launch_relax_match_rs(cos_in, inter, area_p, area_t, score_p, B, N, M, n_valid, m_valid, w_feat,
                                          w_iou, prm, is_test, sim_out, R_out, Rb_out, match_score, det_score,
                                          iters_out, X_final, (hipStream_t)stream);
    const bool exact_ok = (m_valid == nullptr);   // every frame has exactly M templates
    // the tape (dmm_match_train_forward): the taped one-wave kernels write it -- dense batches of <= 16 templates, ragged
    // ones of <= 8; anything else is not taped (the backward re-runs the solver)
    if (tape && taped && !X_final && dmm::relax_tape_bytes(B, N, M, max_iter, proj_iter) > 0 && (exact_ok ? M <= 16 : M <= 8)) {
        uint2 *tape_bits = (uint2 *)tape;
        int *tape_sweeps = (int *)(tape_bits + (size_t)B * max_iter * proj_iter * 64);
        *taped = 1;
#define DMM_TAPED(MT_)                                                                                                   \
    hipLaunchKernelGGL((dmm::relax_match_taped_kernel<MT_, true>), dim3(B), dim3(dmm::solver_block(1, B)), 0,            \
                       (hipStream_t)stream, cos_in, inter, area_p, area_t, score_p, N, M, n_valid, m_valid, w_feat, w_iou, \
                       prm, is_test, sim_out, R_out, Rb_out, match_score, det_score, iters_out, tape_bits, tape_sweeps)
        if (!exact_ok) {
            hipLaunchKernelGGL((dmm::relax_match_ragged_taped_kernel<8>), dim3(B), dim3(dmm::solver_block(1, B)), 0,
                               (hipStream_t)stream, cos_in, inter, area_p, area_t, score_p, N, M, n_valid, m_valid, w_feat,
                               w_iou, prm, is_test, sim_out, R_out, Rb_out, match_score, det_score, iters_out, tape_bits,
                               tape_sweeps);
        } else {
            switch (M) {
                case 1: DMM_TAPED(1); break;   case 2: DMM_TAPED(2); break;   case 3: DMM_TAPED(3); break;
                case 4: DMM_TAPED(4); break;   case 5: DMM_TAPED(5); break;   case 6: DMM_TAPED(6); break;
                case 7: DMM_TAPED(7); break;   case 8: DMM_TAPED(8); break;   case 9: DMM_TAPED(9); break;
                case 10: DMM_TAPED(10); break; case 11: DMM_TAPED(11); break; case 12: DMM_TAPED(12); break;
                case 13: DMM_TAPED(13); break; case 14: DMM_TAPED(14); break; case 15: DMM_TAPED(15); break;
                default: DMM_TAPED(16); break;
            }
        }
#undef DMM_TAPED
        return dmm::check_launch();
    }
    if (!exact_ok && M <= 8 && Pp <= 64) {        // ragged template counts, one wave per frame: per-frame exact bodies
        hipLaunchKernelGGL((dmm::relax_match_ragged_kernel<8>), dim3(B), dim3(dmm::solver_block(1, B)), 0, (hipStream_t)stream, cos_in, inter,
                           area_p, area_t, score_p, N, M, n_valid, m_valid, w_feat, w_iou, prm, is_test, sim_out, R_out,
                           Rb_out, match_score, det_score, iters_out, X_final);
        return dmm::check_launch();
    }
#define DMM_CALL(MT_, NG_, EX_)                                                                                    \
    hipLaunchKernelGGL((dmm::relax_match_kernel<MT_, NG_, EX_>), dim3(B), dim3(dmm::solver_block(NG_, B)), 0, (hipStream_t)stream,   \
                       cos_in, inter, area_p, area_t, score_p, N, M, n_valid, m_valid, w_feat, w_iou, prm, is_test, \
                       sim_out, R_out, Rb_out, match_score, det_score, iters_out, X_final)
    if (clear_tables && cleared && !n_valid && !m_valid) {
        is_test |= dmm::kRelaxClearTables;
        *cleared = 1;
    }
    DMM_DISPATCH_SOLVER(M, Pp, exact_ok, DMM_CALL);
#undef DMM_CALL
    return dmm::check_launch();
}

// (3) for ANY N, M: the general solver (dmm_wide.hip) keeps its state in caller-provided scratch.  Inside the fast kernels'
// envelope this IS dmm_relax_match_f32 (the scratch is not touched); DMM_OPT_FORCE_WIDE (tests) forces the general kernel.
extern "C" size_t dmm_relax_any_scratch_bytes(int B, int N, int M) {
    if (B <= 0 || N <= 0 || M <= 0) return 0;
    return sizeof(float) * (size_t)B * dmm::wide_scratch_floats(M, N > M ? N : M + 1);
}

extern "C" int dmm_relax_match_any_f32(const float *cos_in, const int32_t *inter, const int32_t *area_p,
                                       const int32_t *area_t, const float *score_p, int B, int N, int M,
                                       const int32_t *n_valid, const int32_t *m_valid, float score_weight, int max_iter,
                                       int proj_iter, float lr, int is_test, float *sim_out, float *R_out, float *Rb_out,
                                       float *match_score, float *det_score, int32_t *iters_out, float *X_final,
                                       void *scratch, size_t scratch_bytes, dmm_stream_t stream) {
    if (B < 0 || N < 0 || M < 0 || max_iter < 0 || proj_iter < 0) return DMM_ERR_BAD_ARG;
    if (B == 0 || M == 0) return DMM_OK;
    if (N == 0) return DMM_ERR_BAD_ARG;
    const int Pp = N > M ? N : M + 1;
    if (M <= DMM_MAX_TEMPLATES && Pp <= DMM_MAX_PROPOSALS && dmm::opt(DMM_OPT_FORCE_WIDE) != 1)
        return dmm_relax_match_f32(cos_in, inter, area_p, area_t, score_p, B, N, M, n_valid, m_valid, score_weight, max_iter,
                                   proj_iter, lr, is_test, sim_out, R_out, Rb_out, match_score, det_score, iters_out,
                                   X_final, stream);
    if (!cos_in || !inter || !area_p || !area_t || !score_p || !sim_out || !Rb_out || !match_score || !det_score || !scratch)
        return DMM_ERR_BAD_ARG;
    if (scratch_bytes < dmm_relax_any_scratch_bytes(B, N, M)) return DMM_ERR_WORKSPACE;
    const float w_feat = (float)(1.0 - (double)score_weight);
    return dmm::launch_relax_match_wide(cos_in, inter, area_p, area_t, score_p, B, N, M, n_valid, m_valid, w_feat,
                                        score_weight, dmm::RelaxParams{max_iter, proj_iter, lr}, is_test, sim_out, R_out,
                                        Rb_out, match_score, det_score, iters_out, X_final, (float *)scratch,
                                        (hipStream_t)stream);
}

extern "C" size_t dmm_relax_bwd_workspace_bytes(int B, int N, int M, int max_iter, int proj_iter) {
    if (B <= 0 || N <= 0 || M <= 0 || max_iter < 0 || proj_iter < 0) return 0;
    const int Pp = N > M ? N : M + 1;
    if (M > DMM_MAX_TEMPLATES || Pp > DMM_MAX_PROPOSALS || max_iter > dmm::kMaxTapeOuter ||
        dmm::opt(DMM_OPT_FORCE_WIDE) == 1)                       // the general kernel: all of its state lives here
        return (size_t)B * dmm::wide_bwd_bytes(N, M, max_iter, proj_iter) + 256;
    const size_t threads = 64 * (size_t)((Pp + 63) / 64 == 3 ? 4 : (Pp + 63) / 64);
    return sizeof(uint2) * (size_t)B * (size_t)max_iter * (size_t)proj_iter * threads + 256;
}

extern "C" int dmm_relax_match_bwd_f32(const float *sim, const float *score_p, int B, int N, int M,
                                       const int32_t *n_valid, const int32_t *m_valid, int max_iter, int proj_iter,
                                       float lr, int is_test, const float *dRb, const float *d_match_score,
                                       const float *d_det_score, float *dsim_out, void *workspace,
                                       size_t workspace_bytes, dmm_stream_t stream) {
    return dmm::relax_match_bwd_launch(sim, score_p, B, N, M, n_valid, m_valid, max_iter, proj_iter, lr, is_test, dRb,
                                       d_match_score, d_det_score, dsim_out, workspace, workspace_bytes, nullptr, nullptr,
                                       nullptr, stream);
}

// dmm_relax_match_bwd_f32 proper.  fwd_tape / R_saved / iters_saved (dmm_match_train_backward, all three or none): the tape
// the forward's one-wave kernel kept (relax_tape_bytes > 0 and `taped` returned 1), its R and its iteration counts -- the
// kernel then walks that tape instead of re-running the solver into `workspace`.
int dmm::relax_match_bwd_launch(const float *sim, const float *score_p, int B, int N, int M, const int32_t *n_valid,
                                const int32_t *m_valid, int max_iter, int proj_iter, float lr, int is_test,
                                const float *dRb, const float *d_match_score, const float *d_det_score, float *dsim_out,
                                void *workspace, size_t workspace_bytes, const void *fwd_tape, const float *R_saved,
                                const int32_t *iters_saved, dmm_stream_t stream) {
    if (B < 0 || N < 0 || M < 0 || max_iter < 0 || proj_iter < 0) return DMM_ERR_BAD_ARG;
    if (B == 0 || M == 0) return DMM_OK;
    if (N == 0 || !sim || !score_p || !dsim_out) return DMM_ERR_BAD_ARG;
    const int Pp = N > M ? N : M + 1;
    const bool from_tape = fwd_tape && R_saved && iters_saved && dmm::relax_tape_bytes(B, N, M, max_iter, proj_iter) > 0;
    if (!from_tape && workspace_bytes < dmm_relax_bwd_workspace_bytes(B, N, M, max_iter, proj_iter)) return DMM_ERR_WORKSPACE;
    const dmm::RelaxParams prm{max_iter, proj_iter, lr};
    if (M > DMM_MAX_TEMPLATES || Pp > DMM_MAX_PROPOSALS || max_iter > dmm::kMaxTapeOuter ||
        dmm::opt(DMM_OPT_FORCE_WIDE) == 1) {
        // tables outside the register-resident kernel's envelope (or more outer iterations than its tape index holds):
        // the general backward of dmm_wide.hip -- training at ANY N, M (the reference's autograd is unbounded)
        if (!workspace) return DMM_ERR_BAD_ARG;
        return dmm::launch_relax_match_bwd_wide(sim, score_p, B, N, M, n_valid, m_valid, prm, is_test, dRb, d_match_score,
                                                d_det_score, dsim_out, workspace, (hipStream_t)stream);
    }
    if (!from_tape && !workspace && max_iter * proj_iter > 0) return DMM_ERR_BAD_ARG;
    const bool exact_ok = (m_valid == nullptr);
    uint2 *tape = from_tape ? (uint2 *)const_cast<void *>(fwd_tape) : (uint2 *)workspace;
    const float *Rs = from_tape ? R_saved : nullptr;
    const int32_t *its = from_tape ? iters_saved : nullptr;
    if (!exact_ok && M <= 8 && Pp <= 64) {        // ragged template counts, one wave per frame: per-frame exact bodies
        hipLaunchKernelGGL((dmm::relax_match_bwd_ragged_kernel<8>), dim3(B), dim3(64), 0, (hipStream_t)stream, sim, score_p, N, M,
                           n_valid, m_valid, prm, is_test, dRb, d_match_score, d_det_score, dsim_out, tape, Rs, its);
        return dmm::check_launch();
    }
#define DMM_CALL(MT_, NG_, EX_)                                                                                       \
    hipLaunchKernelGGL((dmm::relax_match_bwd_kernel<MT_, NG_, EX_>), dim3(B), dim3(64 * NG_), 0, (hipStream_t)stream,   \
                       sim, score_p, N, M, n_valid, m_valid, prm, is_test, dRb, d_match_score, d_det_score, dsim_out,  \
                       tape, Rs, its)
    DMM_DISPATCH_SOLVER(M, Pp, exact_ok, DMM_CALL);
#undef DMM_CALL
    return dmm::check_launch();
}

