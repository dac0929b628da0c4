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
// Mapping: one workgroup of NG waves per frame; thread j owns COLUMN j (a proposal) of the
// [M, Pp] problem and keeps its M-row slice of C, X, the three Dykstra increments P0..P2 and the
// running sum of iterates in REGISTERS for all max_iter x proj_iter sweeps -- nothing touches
// memory between the prologue and the epilogue.  Column sums are in-lane, row sums are a 6-step
// DPP tree (+ an LDS hop across waves when Pp > 64).  Elementwise fp32 ops are issued exactly as
// the reference's eager ops (this file is compiled with -ffp-contract=off); the data-dependent
// early exits (relax_match.py:88-89, :96-98) are evaluated on device and the number of executed
// outer iterations is an output.
//
// Roofline: none of HBM / MFMA -- this is a latency-bound dependent chain (~300 VALU ops per
// sweep); throughput comes from running one frame per wave on all 1024 SIMDs.
#include "dmm_common.h"

namespace dmm {

// ---------------------------------------------------------------------------------------------
// Feature row normalisation: out = in / max(||in||, 1e-8)   (first half of cosine_similarity)
// one wave per row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void feature_normalize_kernel(const float *__restrict__ in, int64_t rows, int D,
                                                                float *__restrict__ out, float *__restrict__ norms) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float *x = in + r * D;
    float s = 0.0f;
    for (int d = lane; d < D; d += kWave) s = __builtin_fmaf(x[d], x[d], s);
    float nr = __builtin_sqrtf(wave_sum(s));
    nr = nr > 1e-8f ? nr : 1e-8f;
    for (int d = lane; d < D; d += kWave) out[r * D + d] = x[d] / nr;
    if (norms && lane == 0) norms[r] = nr;
}

// ---------------------------------------------------------------------------------------------
// Cross-wave plumbing for NG > 1 (Pp > 64): per-wave partials go through LDS.
// ---------------------------------------------------------------------------------------------
template <int MT, int NG>
struct BlockRed {
    float *buf;  // [2][NG][MT + 1] floats (double buffered: one barrier per reduction)
    int phase;
    int wave;
    __device__ __forceinline__ BlockRed(float *b, int w) : buf(b), phase(0), wave(w) {}

    // vals[i] are wave-uniform partials; returns block totals (uniform across the block).
    template <int CNT>
    __device__ __forceinline__ void sum(float (&vals)[CNT]) {
        if (NG == 1) return;
        float *p = buf + phase * NG * (MT + 1);
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int i = 0; i < CNT; ++i) p[wave * (MT + 1) + i] = vals[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
            float t = p[i];
#pragma unroll
            for (int w = 1; w < NG; ++w) t = t + p[w * (MT + 1) + i];
            vals[i] = t;
        }
        phase ^= 1;
    }
    __device__ __forceinline__ float max1(float v) {
        if (NG == 1) return v;
        float *p = buf + phase * NG * (MT + 1);
        if ((threadIdx.x & 63) == 0) p[wave * (MT + 1)] = v;
        __syncthreads();
        float t = p[0];
#pragma unroll
        for (int w = 1; w < NG; ++w) { float o = p[w * (MT + 1)]; t = o > t ? o : t; }
        phase ^= 1;
        return t;
    }
    __device__ __forceinline__ float min1(float v) {
        if (NG == 1) return v;
        float *p = buf + phase * NG * (MT + 1);
        if ((threadIdx.x & 63) == 0) p[wave * (MT + 1)] = v;
        __syncthreads();
        float t = p[0];
#pragma unroll
        for (int w = 1; w < NG; ++w) { float o = p[w * (MT + 1)]; t = o < t ? o : t; }
        phase ^= 1;
        return t;
    }
    __device__ __forceinline__ int min1i(int v) {
        if (NG == 1) return v;
        int *p = reinterpret_cast<int *>(buf + phase * NG * (MT + 1));
        if ((threadIdx.x & 63) == 0) p[wave * (MT + 1)] = v;
        __syncthreads();
        int t = p[0];
#pragma unroll
        for (int w = 1; w < NG; ++w) { int o = p[w * (MT + 1)]; t = o < t ? o : t; }
        phase ^= 1;
        return t;
    }
};

struct RelaxParams {
    int max_iter, proj_iter;
    float lr;
};

// ---------------------------------------------------------------------------------------------
// relax_matching core.  C[i] = cost of (row i, this thread's column); n rows, m columns live.
// Threads with col >= m carry zeros everywhere and never change.  On return X[] is the final
// projected iterate, acc[] = sum(X_list); returns len(X_list) - 1.
// ---------------------------------------------------------------------------------------------
template <int MT, int NG>
__device__ __forceinline__ int relax_core(const float (&C)[MT], int n, int m, int col, const RelaxParams prm,
                                          BlockRed<MT, NG> &red, float (&X)[MT], float (&acc)[MT],
                                          float *cost_out /* global [max_iter+1] or null */) {
    const bool live = col < m;
    const float fn = (float)n, fm = (float)m;

    // ---- greedy row-min initialisation (relax_match.py:45-55) ----
    float cmax = -__builtin_inff();
#pragma unroll
    for (int i = 0; i < MT; ++i)
        if (i < n && live) cmax = C[i] > cmax ? C[i] : cmax;
    cmax = red.max1(wave_max(cmax));
    int best_row = 0;
    {
        float bv = C[0];
#pragma unroll
        for (int i = 1; i < MT; ++i)
            if (i < n && C[i] < bv) { bv = C[i]; best_row = i; }   // first argmin over rows
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        X[i] = 0.0f;
        if (i < n) {
            // C_rowmin[i, col]; dead columns are +inf so they never win the row argmin
            float crm = live ? (i == best_row ? C[i] : cmax) : __builtin_inff();
            float vmin = red.min1(wave_min(crm));
            int cand = (live && crm == vmin) ? col : 0x7fffffff;
            int jmin = red.min1i(wave_min_i32(cand));              // first argmin over columns
            if (col == jmin) X[i] = 1.0f;
        }
    }
    float P0[MT], P1[MT], P2[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        P0[i] = 0.0f; P1[i] = 0.0f; P2[i] = 0.0f;
        acc[i] = 0.0f + X[i];                                      // sum(X_list) starts at 0 + X0
    }
    if (cost_out && threadIdx.x == 0) cost_out[0] = 0.0f;

    int len = 1;
    float cost_prev = 0.0f;
    for (int it = 0; it < prm.max_iter; ++it) {
        // gradient step X = X - lr*C  (:69); cost = ||X*C||_F (:70); X_list.append(X) (:71)
        float ss[1] = {0.0f};
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if (i < n) {
                float g = prm.lr * C[i];
                X[i] = X[i] - g;
                float xc = X[i] * C[i];
                ss[0] = __builtin_fmaf(xc, xc, ss[0]);
                acc[i] = acc[i] + X[i];
            }
        }
        ss[0] = wave_sum(ss[0]);
        red.sum(ss);
        const float cost = __builtin_sqrtf(ss[0]);
        if (cost_out && threadIdx.x == 0) cost_out[it + 1] = cost;
        ++len;

        for (int j = 0; j < prm.proj_iter; ++j) {
            float Xs[MT];
            float cs = 0.0f;
            // {X >= 0} (:74-76) then X = Y + P1 and its column sum (:78)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                Xs[i] = X[i];
                if (i < n) {
                    float x = X[i] + P0[i];
                    float y = x > 0.0f ? x : 0.0f;
                    P0[i] = x - y;
                    x = y + P1[i];
                    X[i] = x;
                    cs = cs + x;
                }
            }
            // {column sums <= 1}: project_col (:21-34, :79-80); then X = Y + P2 (:82)
            const bool over = cs > 1.0f;                       // mask = (X_col_sum <= 1)
            const float tc = (cs - 1.0f) / fn;
            float rs[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                rs[i] = 0.0f;
                if (i < n) {
                    float x = X[i];
                    float y = over ? x - tc : x;
                    P1[i] = x - y;
                    x = y + P2[i];
                    X[i] = x;
                    rs[i] = x;
                }
            }
            // {row sums = 1}: project_row (:9-19, :83-84)
#pragma unroll
            for (int i = 0; i < MT; ++i)
                if (i < n) rs[i] = wave_sum(rs[i]);
            red.sum(rs);
            float dd[1] = {0.0f};
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if (i < n && live) {
                    float tr = (rs[i] - 1.0f) / fm;
                    float x = X[i];
                    float y = x - tr;
                    P2[i] = x - y;
                    X[i] = y;                                   // :86
                    float d = y - Xs[i];
                    dd[0] = __builtin_fmaf(d, d, dd[0]);
                }
            }
            // if ||X - X_start|| == 0: break (:88-89)
            dd[0] = wave_sum(dd[0]);
            red.sum(dd);
            if (__builtin_sqrtf(dd[0]) == 0.0f) break;
        }
        if (cost_prev == cost) break;                           // :96-98
        cost_prev = cost;
    }
    return len - 1;
}

// ---------------------------------------------------------------------------------------------
// Full layer kernel: cosine (dot of normalised features) + iou + mix + pad + solver + scores.
// grid = B, block = 64*NG.
// ---------------------------------------------------------------------------------------------
template <int MT, int NG>
__global__ __launch_bounds__(64 * NG) void relax_match_kernel(
    const float *__restrict__ featn_t, const float *__restrict__ featn_p, int D, const int32_t *__restrict__ inter,
    const int32_t *__restrict__ area_p, const int32_t *__restrict__ area_t, const float *__restrict__ score_p, int N,
    int M, const int32_t *__restrict__ n_valid, const int32_t *__restrict__ m_valid, float w_feat, float w_iou,
    RelaxParams prm, int is_test, float *__restrict__ cos_out, float *__restrict__ sim_out, float *__restrict__ R_out,
    float *__restrict__ Rb_out, float *__restrict__ match_score, float *__restrict__ det_score,
    int32_t *__restrict__ iters_out, float *__restrict__ X_final) {
    __shared__ float red_buf[2 * NG * (MT + 1)];
    const int b = blockIdx.x;
    const int col = threadIdx.x;
    BlockRed<MT, NG> red(red_buf, threadIdx.x >> 6);
    const int Nb = n_valid ? n_valid[b] : N;
    const int Mb = m_valid ? m_valid[b] : M;
    const int PpS = N > M ? N : M + 1;                          // table stride
    float *Rb_b = Rb_out + (int64_t)b * M * PpS;
    float *R_b = R_out ? R_out + (int64_t)b * M * PpS : nullptr;
    if (Mb <= 0 || Nb <= 0) {                                   // dead frame: zeros (dmm_model.py:118-122)
        for (int i = threadIdx.x; i < M * PpS; i += 64 * NG) {
            Rb_b[i] = 0.0f;
            if (R_b) R_b[i] = 0.0f;
            if (X_final) X_final[(int64_t)b * M * PpS + i] = 0.0f;
        }
        for (int i = threadIdx.x; i < M * N; i += 64 * NG) {
            sim_out[(int64_t)b * M * N + i] = 0.0f;
            if (cos_out) cos_out[(int64_t)b * M * N + i] = 0.0f;
        }
        for (int i = threadIdx.x; i < M; i += 64 * NG) {
            match_score[(int64_t)b * M + i] = 0.0f;
            det_score[(int64_t)b * M + i] = 0.0f;
        }
        if (iters_out && threadIdx.x == 0) iters_out[b] = 0;
        return;
    }
    const int Pp = Nb > Mb ? Nb : Mb + 1;                       // live solver width (match_model.py:109-113)
    const bool has_prop = col < Nb;

    // ---- cos[m, col] = <tn_m, pn_col> ----
    float cosv[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) cosv[i] = 0.0f;
    {
        const float *k = featn_p + ((int64_t)b * N + (has_prop ? col : 0)) * D;
        const float *q = featn_t + (int64_t)b * M * D;
        int d = 0;
        for (; d + 4 <= D; d += 4) {
            float4u kv = *reinterpret_cast<const float4u *>(k + d);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if (i < Mb) {
                    float4u qv = *reinterpret_cast<const float4u *>(q + (int64_t)i * D + d);
                    float a = cosv[i];
                    a = __builtin_fmaf(qv.x, kv.x, a);
                    a = __builtin_fmaf(qv.y, kv.y, a);
                    a = __builtin_fmaf(qv.z, kv.z, a);
                    a = __builtin_fmaf(qv.w, kv.w, a);
                    cosv[i] = a;
                }
            }
        }
        for (; d < D; ++d) {
            float kv = k[d];
#pragma unroll
            for (int i = 0; i < MT; ++i)
                if (i < Mb) cosv[i] = __builtin_fmaf(q[(int64_t)i * D + d], kv, cosv[i]);
        }
    }

    // ---- sim = (1-w)*cos + w*iou; pad; C = -sim ----
    float C[MT];
    {
        const int32_t *inter_b = inter + (int64_t)b * M * N;
        const int ap = has_prop ? area_p[(int64_t)b * N + col] : 0;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float simv = 0.0f;
            C[i] = 0.0f;
            if (i < Mb && has_prop) {
                const int in = inter_b[(int64_t)i * N + col];
                const int un = ap + area_t[(int64_t)b * M + i] - in;
                const float iou = (float)in / ((float)un + 1e-6f);     // match_helper.py:24-27
                const float a = cosv[i] * w_feat, c = iou * w_iou;
                simv = a + c;                                          // match_model.py:90
                sim_out[(int64_t)b * M * N + (int64_t)i * N + col] = simv;
                if (cos_out) cos_out[(int64_t)b * M * N + (int64_t)i * N + col] = cosv[i];
            }
            if (i < Mb && col < Pp) C[i] = -simv;                      // padded columns: -0.0
        }
    }

    float X[MT], acc[MT];
    const int iters = relax_core<MT, NG>(C, Mb, Pp, col, prm, red, X, acc, nullptr);
    if (iters_out && threadIdx.x == 0) iters_out[b] = iters;

    // ---- R = sum(X_list)/len; logic; Rb; scores ----
    const float flen = (float)(iters + 1);
    const float sc = has_prop ? score_p[(int64_t)b * N + col] : 0.0f;
    const bool livec = col < Pp;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        if (i < Mb) {
            const float r = acc[i] / flen;                             // match_model.py:121
            const float rmax = red.max1(wave_max(livec ? r : -__builtin_inff()));
            const float lg = is_test ? (r == rmax ? 1.0f : 0.0f) : (r > 0.01f ? 1.0f : 0.0f);
            const float rb = livec ? r * lg : 0.0f;                    // :130
            const float rc = r < 0.0f ? 0.0f : (r > 1.0f ? 1.0f : r);
            const float ms = red.max1(wave_max(livec ? rc * (-C[i]) : -__builtin_inff()));   // :146
            float ds[1] = {wave_sum(sc * rb)};                         // :147
            red.sum(ds);
            if (col < PpS) {
                Rb_b[(int64_t)i * PpS + col] = rb;
                if (R_b) R_b[(int64_t)i * PpS + col] = livec ? r : 0.0f;
                if (X_final) X_final[(int64_t)b * M * PpS + (int64_t)i * PpS + col] = livec ? X[i] : 0.0f;
            }
            if (threadIdx.x == 0) {
                match_score[(int64_t)b * M + i] = ms;
                det_score[(int64_t)b * M + i] = ds[0];
            }
        }
    }
    // rows of dead templates: zeros
    for (int i = Mb; i < M; ++i) {
        if (col < PpS) {
            Rb_b[(int64_t)i * PpS + col] = 0.0f;
            if (R_b) R_b[(int64_t)i * PpS + col] = 0.0f;
            if (X_final) X_final[(int64_t)b * M * PpS + (int64_t)i * PpS + col] = 0.0f;
        }
        if (col < N) {
            sim_out[(int64_t)b * M * N + (int64_t)i * N + col] = 0.0f;
            if (cos_out) cos_out[(int64_t)b * M * N + (int64_t)i * N + col] = 0.0f;
        }
        if (threadIdx.x == 0) {
            match_score[(int64_t)b * M + i] = 0.0f;
            det_score[(int64_t)b * M + i] = 0.0f;
        }
    }
    // live rows, dead proposal columns of sim/cos: zeros
    if (!has_prop && col < N) {
        for (int i = 0; i < Mb; ++i) {
            sim_out[(int64_t)b * M * N + (int64_t)i * N + col] = 0.0f;
            if (cos_out) cos_out[(int64_t)b * M * N + (int64_t)i * N + col] = 0.0f;
        }
    }
}

// Solver-only kernel on a caller-provided C [B, n, m].
template <int MT, int NG>
__global__ __launch_bounds__(64 * NG) void relax_solve_kernel(const float *__restrict__ Cin, int n, int m,
                                                              RelaxParams prm, float *__restrict__ X_final,
                                                              float *__restrict__ R_out, float *__restrict__ cost_out,
                                                              int32_t *__restrict__ iters_out) {
    __shared__ float red_buf[2 * NG * (MT + 1)];
    const int b = blockIdx.x, col = threadIdx.x;
    BlockRed<MT, NG> red(red_buf, threadIdx.x >> 6);
    float C[MT], X[MT], acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) C[i] = (i < n && col < m) ? Cin[((int64_t)b * n + i) * m + col] : 0.0f;
    const int iters = relax_core<MT, NG>(C, n, m, col, prm, red, X, acc,
                                         cost_out ? cost_out + (int64_t)b * (prm.max_iter + 1) : nullptr);
    const float flen = (float)(iters + 1);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        if (i < n && col < m) {
            if (X_final) X_final[((int64_t)b * n + i) * m + col] = X[i];
            if (R_out) R_out[((int64_t)b * n + i) * m + col] = acc[i] / flen;
        }
    }
    if (iters_out && threadIdx.x == 0) iters_out[b] = iters;
}

}  // namespace dmm

extern "C" int dmm_feature_normalize_f32(const float *in, int64_t rows, int D, float *out, float *norms,
                                         dmm_stream_t stream) {
    if (rows < 0 || D < 0) return DMM_ERR_BAD_ARG;
    if (rows == 0 || D == 0) return DMM_OK;
    if (!in || !out) return DMM_ERR_BAD_ARG;
    const int64_t blocks = (rows + 3) / 4;
    hipLaunchKernelGGL(dmm::feature_normalize_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in,
                       rows, D, out, norms);
    return dmm::check_launch();
}

#define DMM_DISPATCH_MT_NG(M_, W_, CALL)                    \
    do {                                                    \
        const int ng_ = ((W_) + 63) / 64;                   \
        if ((M_) <= 4) {                                    \
            if (ng_ <= 1) { CALL(4, 1); }                   \
            else if (ng_ == 2) { CALL(4, 2); }              \
            else { CALL(4, 4); }                            \
        } else if ((M_) <= 8) {                             \
            if (ng_ <= 1) { CALL(8, 1); }                   \
            else if (ng_ == 2) { CALL(8, 2); }              \
            else { CALL(8, 4); }                            \
        } else if ((M_) <= 16) {                            \
            if (ng_ <= 1) { CALL(16, 1); }                  \
            else if (ng_ == 2) { CALL(16, 2); }             \
            else { CALL(16, 4); }                           \
        } else {                                            \
            if (ng_ <= 1) { CALL(32, 1); }                  \
            else if (ng_ == 2) { CALL(32, 2); }             \
            else { CALL(32, 4); }                           \
        }                                                   \
    } while (0)

extern "C" int dmm_relax_match_f32(const float *featn_t, const float *featn_p, int D, const int32_t *inter,
                                   const int32_t *area_p, const int32_t *area_t, const float *score_p, int B, int N,
                                   int M, const int32_t *n_valid, const int32_t *m_valid, float score_weight,
                                   int max_iter, int proj_iter, float lr, int is_test, float *cos_out, float *sim_out,
                                   float *R_out, float *Rb_out, float *match_score, float *det_score,
                                   int32_t *iters_out, float *X_final, dmm_stream_t stream) {
    if (B < 0 || N < 0 || M < 0 || D < 0 || max_iter < 0 || proj_iter < 0) return DMM_ERR_BAD_ARG;
    if (B == 0 || M == 0) return DMM_OK;
    if (N == 0) return DMM_ERR_BAD_ARG;
    if (!featn_t || !featn_p || !inter || !area_p || !area_t || !score_p || !sim_out || !Rb_out || !match_score ||
        !det_score)
        return DMM_ERR_BAD_ARG;
    const int Pp = N > M ? N : M + 1;
    if (M > DMM_MAX_TEMPLATES || Pp > DMM_MAX_PROPOSALS) return DMM_ERR_UNSUPPORTED;
    const dmm::RelaxParams prm{max_iter, proj_iter, lr};
    // python: sim*(1-w) + iou*w with w a python float -> both scalars rounded to fp32 once
    const float w_feat = (float)(1.0 - (double)score_weight), w_iou = score_weight;
#define DMM_CALL(MT_, NG_)                                                                                           \
    hipLaunchKernelGGL((dmm::relax_match_kernel<MT_, NG_>), dim3(B), dim3(64 * NG_), 0, (hipStream_t)stream, featn_t, \
                       featn_p, D, inter, area_p, area_t, score_p, N, M, n_valid, m_valid, w_feat, w_iou, prm,       \
                       is_test, cos_out, sim_out, R_out, Rb_out, match_score, det_score, iters_out, X_final)
    DMM_DISPATCH_MT_NG(M, Pp, DMM_CALL);
#undef DMM_CALL
    return dmm::check_launch();
}

extern "C" int dmm_relax_solve_f32(const float *C, int B, int n, int m, int max_iter, int proj_iter, float lr,
                                   float *X_final, float *R_out, float *cost_out, int32_t *iters_out,
                                   dmm_stream_t stream) {
    if (B < 0 || n <= 0 || m <= 0 || max_iter < 0 || proj_iter < 0) return DMM_ERR_BAD_ARG;
    if (B == 0) return DMM_OK;
    if (!C) return DMM_ERR_BAD_ARG;
    if (n > DMM_MAX_TEMPLATES || m > DMM_MAX_PROPOSALS) return DMM_ERR_UNSUPPORTED;
    const dmm::RelaxParams prm{max_iter, proj_iter, lr};
#define DMM_CALL(MT_, NG_)                                                                                        \
    hipLaunchKernelGGL((dmm::relax_solve_kernel<MT_, NG_>), dim3(B), dim3(64 * NG_), 0, (hipStream_t)stream, C, n, \
                       m, prm, X_final, R_out, cost_out, iters_out)
    DMM_DISPATCH_MT_NG(n, m, DMM_CALL);
#undef DMM_CALL
    return dmm::check_launch();
}
