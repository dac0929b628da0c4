// dmm_wide.hip -- the layer's GENERAL forms, for tables outside the envelope the fast kernels are compiled for
// (M <= 32 template rows, solver width Pp = max(N, M + 1) <= 256): feature similarity and the relaxed-assignment solver with
// its prologue / epilogue for ANY N and M (the general mask mix sits next to the fast one in dmm_mix.hip).
//
// The reference is unbounded (relax_matching, dmm/modules/submodules/relax_match.py:36-105, takes any [n, m] cost matrix;
// get_cosine_score, dmm/utils/match_helper.py:51-64; match_with_first_frame, dmm/modules/match_model.py:98-148); DMM-Net's
// own configurations (<= 100 proposals, a handful of objects) never leave the envelope, so these kernels are written for
// CORRECTNESS, not speed: the solver state lives in a global-memory scratch (L2 resident), one workgroup per frame walks
// the reference's steps with a barrier between them.  They issue the same fp32 operations in the same order as the fast
// kernels -- ATen's reduction orders from dmm_torch_order.h, which are written for any length -- so results are bit
// identical to the reference's CPU path here too (tests/test_gpu_wide.py: against the oracle at wide shapes, and against
// every layer golden with DMM_OPT_FORCE_WIDE forcing these kernels inside the envelope).
//
// Roofline: none of it is bound by HBM or MFMA -- dependent chains through L2; the IoU counts (dmm_cost.hip tiles any
// N x M) remain the streaming part.
#include "dmm_solve.h"

namespace dmm {

constexpr int kWideThreads = 256;          // similarity kernel
constexpr int kWideSolverThreads = 1024;   // solver: one workgroup per frame, 16 waves share its element-wise passes

// ---------------------------------------------------------------------------------------------
// cos[b, m, n] = sum_d RN(tn[m, d] * pn[n, d]) over the [D, Nb] slab of products in ATen's outer-sum order (columns below
// outer_class_bound(Nb): one cascade chain; the rest: ILP-4 row_sum), or -- one live proposal -- the inner sum over D.
// grid = (ceil(M * N / 256) | ceil(M * 8 / 256), B).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWideThreads) void cosine_wide_kernel(const float *__restrict__ tn,
                                                                   const float *__restrict__ pn, int N, int M, int D,
                                                                   const int32_t *__restrict__ n_valid,
                                                                   const int32_t *__restrict__ m_valid,
                                                                   float *__restrict__ cos_out) {
    const int b = blockIdx.y;
    const int Nb = n_valid ? n_valid[b] : N, Mb = m_valid ? m_valid[b] : M;
    float *out = cos_out + (int64_t)b * M * N;
    const int64_t gid = (int64_t)blockIdx.x * kWideThreads + threadIdx.x;
    if (Nb == 1) {                                        // [O, D, 1]: the reduced dimension is the fastest one
        const int m = (int)(gid >> 3), l = threadIdx.x & 7;
        const bool live = m < Mb;
        const float *q = tn + ((int64_t)b * M + (live ? m : 0)) * D, *k = pn + (int64_t)b * N * D;
        const float s = torder::inner_sum_group8(D, l, [&](long d) { return q[d] * k[d]; });
        if (l == 0 && m < M) out[(int64_t)m * N] = live ? s : 0.0f;
        if (m < M)
            for (int n = 1 + l; n < N; n += 8) out[(int64_t)m * N + n] = 0.0f;
        return;
    }
    if (gid >= (int64_t)M * N) return;
    const int m = (int)(gid / N), n = (int)(gid - (int64_t)m * N);
    if (m >= Mb || n >= Nb) { out[gid] = 0.0f; return; }
    const float *q = tn + ((int64_t)b * M + m) * D, *k = pn + ((int64_t)b * N + n) * D;
    out[gid] = torder::outer_sum_col(D, n < torder::outer_class_bound(Nb), [&](long d) { return q[d] * k[d]; });
}

int launch_cosine_wide(const float *featn_t, const float *featn_p, int B, int N, int M, int D, const int32_t *n_valid,
                       const int32_t *m_valid, float *cos_out, hipStream_t stream) {
    const int64_t per = (int64_t)M * (N > 8 ? N : 8);     // 8 lanes per output when a frame has ONE live proposal
    const int64_t gx = (per + kWideThreads - 1) / kWideThreads;
    if (gx > 0x7fffffffLL) return DMM_ERR_UNSUPPORTED;
    for (int b0 = 0; b0 < B; b0 += 65535) {
        const int nb = B - b0 < 65535 ? B - b0 : 65535;
        hipLaunchKernelGGL(cosine_wide_kernel, dim3((unsigned)gx, nb), dim3(kWideThreads), 0, stream,
                           featn_t + (int64_t)b0 * M * D, featn_p + (int64_t)b0 * N * D, N, M, D,
                           n_valid ? n_valid + b0 : nullptr, m_valid ? m_valid + b0 : nullptr,
                           cos_out + (int64_t)b0 * M * N);
    }
    return check_launch();
}

// ---------------------------------------------------------------------------------------------
// Solver + prologue + epilogue, one workgroup per frame, state in `scratch` (wide_scratch_floats(M, Pp) floats per frame).
// Same contract as relax_match_kernel (dmm_solve.hip): sim, R, Rb, match_score, det_score, iters, X_final.
// ---------------------------------------------------------------------------------------------
size_t wide_scratch_floats(int M, int PpS) { return (size_t)9 * M * PpS + (size_t)PpS + 2 * (size_t)M + 64; }

template <int THREADS = kWideSolverThreads>
__device__ __forceinline__ float wide_block_max(float v, float *sh) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = sh[0];
#pragma unroll
    for (int k = 1; k < THREADS / 64; ++k) r = sh[k] > r ? sh[k] : r;
    return r;
}
// max over the aligned 8-lane group (every lane gets it)
__device__ __forceinline__ float group8_max(float v) {
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) {
        const float o = __shfl_xor(v, d, 8);
        v = o > v ? o : v;
    }
    return v;
}

__global__ __launch_bounds__(kWideSolverThreads) void relax_match_wide_kernel(
    const float *__restrict__ cos_in, const int32_t *__restrict__ inter, const int32_t *__restrict__ area_p,
    const int32_t *__restrict__ area_t, const float *__restrict__ score_p, int N, int M,
    const int32_t *__restrict__ n_valid, const int32_t *__restrict__ m_valid, float w_feat, float w_iou, RelaxParams prm,
    int is_test, float *__restrict__ sim_out, float *__restrict__ R_out, float *__restrict__ Rb_out,
    float *__restrict__ match_score, float *__restrict__ det_score, int32_t *__restrict__ iters_out,
    float *__restrict__ X_final, float *__restrict__ scratch, int64_t scratch_stride, int x_in_lds) {
    __shared__ float sh[kWideSolverThreads / 64 + 1];
    extern __shared__ __attribute__((aligned(16))) float x_lds[];     // the iterate X when the table fits (the sums read it)
    const int b = blockIdx.x, tid = threadIdx.x, l = tid & 7, grp = tid >> 3;
    constexpr int NT = kWideSolverThreads, NGRP = kWideSolverThreads / 8;
    const int Nb = n_valid ? n_valid[b] : N;
    const int Mb = m_valid ? m_valid[b] : M;
    const int PpS = N > M ? N : M + 1;                                 // table stride
    float *Rb_b = Rb_out + (int64_t)b * M * PpS;
    float *R_b = R_out ? R_out + (int64_t)b * M * PpS : nullptr;
    float *X_b = X_final ? X_final + (int64_t)b * M * PpS : nullptr;
    float *sim_b = sim_out + (int64_t)b * M * N;
    // everything the live block below does not write: zeros (dead frame: all of it, dmm_model.py:118-122)
    const bool dead = Mb <= 0 || Nb <= 0;
    const int n = dead ? 0 : Mb;                                      // live rows
    const int m = dead ? 0 : (Nb > Mb ? Nb : Mb + 1);                 // live solver width (match_model.py:109-113)
    for (int e = tid; e < M * PpS; e += NT) {
        const int i = e / PpS, c = e - i * PpS;
        if (i >= n || c >= m) {
            Rb_b[e] = 0.0f;
            if (R_b) R_b[e] = 0.0f;
            if (X_b) X_b[e] = 0.0f;
        }
    }
    for (int e = tid; e < M * N; e += NT) {
        const int i = e / N, c = e - i * N;
        if (i >= n || c >= Nb) sim_b[e] = 0.0f;
    }
    for (int i = n + tid; i < M; i += NT) {
        match_score[(int64_t)b * M + i] = 0.0f;
        det_score[(int64_t)b * M + i] = 0.0f;
    }
    if (dead) {
        if (iters_out && tid == 0) iters_out[b] = 0;
        return;
    }
    const int cnt = n * m;
    float *C = scratch + (int64_t)b * scratch_stride;
    const size_t cap = (size_t)M * PpS;
    float *X = x_in_lds ? x_lds : C + cap, *Y = C + 2 * cap, *P0 = Y + cap, *P1 = P0 + cap, *P2 = P1 + cap, *Xs = P2 + cap, *acc = Xs + cap,
          *tmp = acc + cap, *tc = tmp + cap, *rt = tc + PpS;
    int *idx = reinterpret_cast<int *>(rt + M);

    // ---- sim = (1-w)*cos + w*iou (match_model.py:90, match_helper.py:24-27); pad; C = -sim ----
    {
        const float *cos_b = cos_in + (int64_t)b * M * N;
        const int32_t *inter_b = inter + (int64_t)b * M * N;
        for (int e = tid; e < cnt; e += NT) {
            const int i = e / m, c = e - i * m;
            float simv = 0.0f;
            if (c < Nb) {
                const int in = inter_b[(int64_t)i * N + c];
                const int un = area_p[(int64_t)b * N + c] + area_t[(int64_t)b * M + i] - in;
                const float iou = (float)in / ((float)un + 1e-6f);
                const float a = cos_b[(int64_t)i * N + c] * w_feat, cc = iou * w_iou;
                simv = a + cc;
                sim_b[(int64_t)i * N + c] = simv;
            }
            C[e] = -simv;                                              // padded columns: -0.0
        }
    }
    __syncthreads();

    // ---- greedy row-min initialisation (relax_match.py:45-55); max / first-argmin are order free ----
    {
        float cm = -__builtin_inff();
        for (int e = tid; e < cnt; e += NT) cm = C[e] > cm ? C[e] : cm;
        const float cmax = wide_block_max(cm, sh);
        for (int c = tid; c < m; c += NT) {
            int best = 0;
            float bv = C[c];
            for (int i = 1; i < n; ++i)
                if (C[i * m + c] < bv) { bv = C[i * m + c]; best = i; }   // first argmin over rows
            for (int i = 0; i < n; ++i) Y[i * m + c] = i == best ? C[i * m + c] : cmax;
        }
        __syncthreads();
        for (int i = tid; i < n; i += NT) {
            int best = 0;
            float bv = Y[i * m];
            for (int c = 1; c < m; ++c)
                if (Y[i * m + c] < bv) { bv = Y[i * m + c]; best = c; }   // first argmin over columns
            idx[i] = best;
        }
        __syncthreads();
        for (int e = tid; e < cnt; e += NT) {
            const int i = e / m, c = e - i * m;
            const float x0 = c == idx[i] ? 1.0f : 0.0f;
            X[e] = x0;
            acc[e] = 0.0f + x0;                                        // sum(X_list) starts at 0 + X0
            P0[e] = 0.0f; P1[e] = 0.0f; P2[e] = 0.0f;
        }
    }
    __syncthreads();

    const float fn = (float)n, fm = (float)m;
    const int cbound = torder::outer_class_bound(m);
    int len = 1;
    float cost_prev = 0.0f;
    for (int it = 0; it < prm.max_iter; ++it) {
        // gradient step X = X - lr*C (:69); cost = ||X*C||_F (:70); X_list.append(X) (:71)
        for (int e = tid; e < cnt; e += NT) {
            const float g = prm.lr * C[e];
            const float x = X[e] - g;
            X[e] = x;
            tmp[e] = x * C[e];
            acc[e] = acc[e] + x;
        }
        __syncthreads();
        if (tid < 8) {
            const float c = torder::norm2_group8(cnt, tid, [&](long i) { return tmp[i]; });
            if (tid == 0) sh[kWideSolverThreads / 64] = c;
        }
        __syncthreads();
        const float cost = sh[kWideSolverThreads / 64];
        ++len;
        for (int j = 0; j < prm.proj_iter; ++j) {
            // {X >= 0} (:74-76), then X = Y + P1 (:78)
            for (int e = tid; e < cnt; e += NT) {
                const float xs = X[e];
                Xs[e] = xs;
                const float x = xs + P0[e];
                const float y = x > 0.0f ? x : 0.0f;
                P0[e] = x - y;
                X[e] = y + P1[e];
            }
            __syncthreads();
            // project_col (:21-34): column sums in ATen's outer-sum order
            for (int c = tid; c < m; c += NT) {
                const float cs = torder::outer_sum_col(n, c < cbound, [&](long i) { return X[i * m + c]; });
                tc[c] = cs <= 1.0f ? 0.0f : (cs - 1.0f) / fn;          // (x - 0 = x exactly; a NaN sum gives a NaN step)
            }
            __syncthreads();
            for (int e = tid; e < cnt; e += NT) {
                const int c = e % m;
                const float x = X[e];
                const float y = x - tc[c];
                P1[e] = x - y;
                X[e] = y + P2[e];                                      // (:82)
            }
            __syncthreads();
            // project_row (:9-19): row sums in ATen's inner-sum order, one 8-lane group per row
            for (int i = grp; i < n; i += NGRP) {
                const float s = torder::inner_sum_group8(m, l, [&](long k) { return X[i * m + k]; });
                if (l == 0) rt[i] = (s - 1.0f) / fm;
            }
            __syncthreads();
            int moved = 0;
            for (int e = tid; e < cnt; e += NT) {
                const int i = e / m;
                const float x = X[e];
                const float y = x - rt[i];
                P2[e] = x - y;
                X[e] = y;                                              // (:86)
                const float d = y - Xs[e];
                const float sq = d * d;
                moved |= !(sq == 0.0f);                                // ||X - X_start|| == 0 (:88): every square is zero
            }
            if (!__syncthreads_or(moved)) break;
        }
        if (cost_prev == cost) break;                                  // (:96-98)
        cost_prev = cost;
    }
    const int iters = len - 1;
    if (iters_out && tid == 0) iters_out[b] = iters;

    // ---- R = sum(X_list)/len; logic; Rb; scores (match_model.py:121-147), one 8-lane group per row ----
    const float flen = (float)len;
    for (int i = grp; i < n; i += NGRP) {
        float mx = -__builtin_inff();
        for (int c = l; c < m; c += 8) {
            const float r = acc[i * m + c] / flen;
            mx = r > mx ? r : mx;
        }
        mx = group8_max(mx);
        auto rb_of = [&](int c) {
            const float r = acc[i * m + c] / flen;
            const float lg = is_test ? (r == mx ? 1.0f : 0.0f) : (r > 0.01f ? 1.0f : 0.0f);
            return r * lg;                                             // (:130)
        };
        float ms = -__builtin_inff();
        for (int c = l; c < m; c += 8) {
            const float r = acc[i * m + c] / flen;
            const float rb = rb_of(c);
            const float rc = r < 0.0f ? 0.0f : (r > 1.0f ? 1.0f : r);
            const float v = rc * (-C[i * m + c]);                      // (:146)
            ms = v > ms ? v : ms;
            Rb_b[(int64_t)i * PpS + c] = rb;
            if (R_b) R_b[(int64_t)i * PpS + c] = r;
            if (X_b) X_b[(int64_t)i * PpS + c] = X[i * m + c];
        }
        ms = group8_max(ms);
        const float ds = torder::inner_sum_group8(m, l, [&](long k) {     // (score * Rb).sum(1) (:147)
            const float sc = k < Nb ? score_p[(int64_t)b * N + k] : 0.0f;
            return sc * rb_of((int)k);
        });
        if (l == 0) {
            match_score[(int64_t)b * M + i] = ms;
            det_score[(int64_t)b * M + i] = ds;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same solver with its state in REGISTERS (the L2-resident form above spends its time in latency:
// 12 dependent global round trips per element-wise pass and thread, 2.8 ms for 300 x 40 at 20 x 5).  Thread t owns elements
// t, t + 512, ... (K of them, K = ceil(M * Pp / 512) <= 24) and keeps their C, X, the three Dykstra increments, the
// sweep's start copy and the sum of iterates in registers for the whole solve; only what OTHER threads read goes through
// LDS: the iterate X (column / row sums) and the products X * C (cost norm) -- two M x Pp tables.  Same operations in the
// same order as relax_match_wide_kernel (element-wise steps are order free; the sums are the same torder:: routines on the
// same values), so the results stay bit identical.  Tables of more than 12288 entries keep the kernel above.
// 512 threads per frame: 8 waves = 2 per SIMD = 256 VGPRs (1024 threads would cap at 128).  Measured (round 4, LABLOG;
// one frame, 20 x 5): 50 x 40 solver 0.49 ms, 300 x 40 1.29 ms (L2-resident form: 0.95 / 2.5): the K = 24 instantiation still
// spills ~150 registers inside the sweep (~100 registers of fixed cost from the inlined ATen-order sums + ~13 per element).
// ---------------------------------------------------------------------------------------------
constexpr int kWideRegThreads = 512;
template <int K>
__global__ __launch_bounds__(kWideRegThreads) void relax_match_wide_reg_kernel(
    const float *__restrict__ cos_in, const int32_t *__restrict__ inter, const int32_t *__restrict__ area_p,
    const int32_t *__restrict__ area_t, const float *__restrict__ score_p, int N, int M,
    const int32_t *__restrict__ n_valid, const int32_t *__restrict__ m_valid, float w_feat, float w_iou, RelaxParams prm,
    int is_test, float *__restrict__ sim_out, float *__restrict__ R_out, float *__restrict__ Rb_out,
    float *__restrict__ match_score, float *__restrict__ det_score, int32_t *__restrict__ iters_out,
    float *__restrict__ X_final, float *__restrict__ scratch, int64_t scratch_stride) {
    __shared__ float sh[kWideRegThreads / 64 + 1];
    extern __shared__ __attribute__((aligned(16))) float lds_t[];      // XL [cap] | PL [cap] | SL [cap] | tc [PpS] | rt [M] | idx [M]
    const int b = blockIdx.x, tid = threadIdx.x, l = tid & 7, grp = tid >> 3;
    constexpr int NT = kWideRegThreads, NGRP = kWideRegThreads / 8;
    const int Nb = n_valid ? n_valid[b] : N;
    const int Mb = m_valid ? m_valid[b] : M;
    const int PpS = N > M ? N : M + 1;
    float *Rb_b = Rb_out + (int64_t)b * M * PpS;
    float *R_b = R_out ? R_out + (int64_t)b * M * PpS : nullptr;
    float *X_b = X_final ? X_final + (int64_t)b * M * PpS : nullptr;
    float *sim_b = sim_out + (int64_t)b * M * N;
    const bool dead = Mb <= 0 || Nb <= 0;
    const int n = dead ? 0 : Mb;
    const int m = dead ? 0 : (Nb > Mb ? Nb : Mb + 1);
    for (int e = tid; e < M * PpS; e += NT) {
        const int i = e / PpS, c = e - i * PpS;
        if (i >= n || c >= m) {
            Rb_b[e] = 0.0f;
            if (R_b) R_b[e] = 0.0f;
            if (X_b) X_b[e] = 0.0f;
        }
    }
    for (int e = tid; e < M * N; e += NT) {
        const int i = e / N, c = e - i * N;
        if (i >= n || c >= Nb) sim_b[e] = 0.0f;
    }
    for (int i = n + tid; i < M; i += NT) {
        match_score[(int64_t)b * M + i] = 0.0f;
        det_score[(int64_t)b * M + i] = 0.0f;
    }
    if (dead) {
        if (iters_out && tid == 0) iters_out[b] = 0;
        return;
    }
    const int cnt = n * m;
    const size_t cap = (size_t)M * PpS;
    // XL: the iterate (what the sums read); PL: X * C for the cost norm; SL: the sweep's start copy (read back once per
    // sweep by its owner -- keeping it in registers cost 1/7 of the register budget); the sum of iterates is touched once
    // per OUTER iteration and lives in the caller's scratch (L2)
    float *XL = lds_t, *PL = XL + cap, *SL = PL + cap, *tc = SL + cap, *rt = tc + PpS;
    int *idx = reinterpret_cast<int *>(rt + M);
    // ... and so does the cost C (read once per outer iteration: the gradient step and the cost norm's products)
    float *Cg = scratch + (int64_t)b * scratch_stride, *accg = Cg + cap;

    float Xr[K], P0[K], P1[K], P2[K];
    int ic[K];                                                         // (row << 16) | column of element k
    bool on[K];
    // ---- sim = (1-w)*cos + w*iou (match_model.py:90, match_helper.py:24-27); pad; C = -sim ----
    {
        const float *cos_b = cos_in + (int64_t)b * M * N;
        const int32_t *inter_b = inter + (int64_t)b * M * N;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int e = tid + k * NT;
            on[k] = e < cnt;
            const int i = on[k] ? e / m : 0, c = on[k] ? e - i * m : 0;
            ic[k] = (i << 16) | c;
            float simv = 0.0f;
            if (on[k] && c < Nb) {
                const int in = inter_b[(int64_t)i * N + c];
                const int un = area_p[(int64_t)b * N + c] + area_t[(int64_t)b * M + i] - in;
                const float iou = (float)in / ((float)un + 1e-6f);
                const float a = cos_b[(int64_t)i * N + c] * w_feat, cc = iou * w_iou;
                simv = a + cc;
                sim_b[(int64_t)i * N + c] = simv;
            }
            if (on[k]) {
                XL[e] = -simv;                                         // padded columns: -0.0
                Cg[e] = -simv;
            }
        }
    }
    __syncthreads();
    // ---- greedy row-min initialisation (relax_match.py:45-55) on the LDS copy of C; max / first-argmin are order free ----
    {
        float cm = -__builtin_inff();
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (on[k]) cm = XL[tid + k * NT] > cm ? XL[tid + k * NT] : cm;
        const float cmax = wide_block_max<kWideRegThreads>(cm, sh);
        for (int c = tid; c < m; c += NT) {
            int best = 0;
            float bv = XL[c];
            for (int i = 1; i < n; ++i)
                if (XL[i * m + c] < bv) { bv = XL[i * m + c]; best = i; }
            for (int i = 0; i < n; ++i) PL[i * m + c] = i == best ? XL[i * m + c] : cmax;
        }
        __syncthreads();
        for (int i = tid; i < n; i += NT) {
            int best = 0;
            float bv = PL[i * m];
            for (int c = 1; c < m; ++c)
                if (PL[i * m + c] < bv) { bv = PL[i * m + c]; best = c; }
            idx[i] = best;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float x0 = (on[k] && (ic[k] & 0xffff) == idx[ic[k] >> 16]) ? 1.0f : 0.0f;
            Xr[k] = x0;
            if (on[k]) accg[tid + k * NT] = 0.0f + x0;
            P0[k] = 0.0f; P1[k] = 0.0f; P2[k] = 0.0f;
        }
    }
    __syncthreads();

    const float fn = (float)n, fm = (float)m;
    const int cbound = torder::outer_class_bound(m);
    int len = 1;
    float cost_prev = 0.0f;
    for (int it = 0; it < prm.max_iter; ++it) {
        // gradient step X = X - lr*C (:69); cost = ||X*C||_F (:70); X_list.append(X) (:71)
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float cv = on[k] ? Cg[tid + k * NT] : 0.0f;
            const float g = prm.lr * cv;
            const float x = Xr[k] - g;
            Xr[k] = x;
            if (on[k]) {
                PL[tid + k * NT] = x * cv;
                accg[tid + k * NT] = accg[tid + k * NT] + x;
            }
        }
        __syncthreads();
        if (tid < 8) {
            const float c = torder::norm2_group8(cnt, tid, [&](long i) { return PL[i]; });
            if (tid == 0) sh[kWideRegThreads / 64] = c;
        }
        __syncthreads();
        const float cost = sh[kWideRegThreads / 64];
        ++len;
        for (int j = 0; j < prm.proj_iter; ++j) {
            // {X >= 0} (:74-76), then X = Y + P1 (:78)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float xs = Xr[k];
                if (on[k]) SL[tid + k * NT] = xs;
                const float x = xs + P0[k];
                const float y = x > 0.0f ? x : 0.0f;
                P0[k] = x - y;
                Xr[k] = y + P1[k];
                if (on[k]) XL[tid + k * NT] = Xr[k];
            }
            __syncthreads();
            // project_col (:21-34): column sums in ATen's outer-sum order
            for (int c = tid; c < m; c += NT) {
                const float cs = torder::outer_sum_col_batched(n, c < cbound, [&](int i) { return XL[i * m + c]; });
                tc[c] = cs <= 1.0f ? 0.0f : (cs - 1.0f) / fn;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float x = Xr[k];
                const float y = x - (on[k] ? tc[ic[k] & 0xffff] : 0.0f);
                P1[k] = x - y;
                Xr[k] = y + P2[k];                                     // (:82)
                if (on[k]) XL[tid + k * NT] = Xr[k];
            }
            __syncthreads();
            // project_row (:9-19): row sums in ATen's inner-sum order, one 8-lane group per row
            for (int i = grp; i < n; i += NGRP) {
                const float s = m < 512 ? torder::inner_sum_group8_batched(m, l, [&](int q) { return XL[i * m + q]; })
                                        : torder::inner_sum_group8(m, l, [&](long q) { return XL[i * m + q]; });
                if (l == 0) rt[i] = (s - 1.0f) / fm;
            }
            __syncthreads();
            int moved = 0;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float x = Xr[k];
                const float y = x - (on[k] ? rt[ic[k] >> 16] : 0.0f);
                P2[k] = x - y;
                Xr[k] = y;                                             // (:86)
                const float d = y - (on[k] ? SL[tid + k * NT] : y);
                const float sq = d * d;
                moved |= on[k] && !(sq == 0.0f);                       // ||X - X_start|| == 0 (:88): every square is zero
            }
            if (!__syncthreads_or(moved)) break;
        }
        if (cost_prev == cost) break;                                  // (:96-98)
        cost_prev = cost;
    }
    const int iters = len - 1;
    if (iters_out && tid == 0) iters_out[b] = iters;

    // ---- R = sum(X_list)/len; logic; Rb; scores (match_model.py:121-147): acc -> XL, C -> PL, X (if asked) straight out ----
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (on[k]) {
            XL[tid + k * NT] = accg[tid + k * NT];
            PL[tid + k * NT] = Cg[tid + k * NT];
            if (X_b) X_b[(int64_t)(ic[k] >> 16) * PpS + (ic[k] & 0xffff)] = Xr[k];
        }
    }
    __syncthreads();
    const float flen = (float)len;
    for (int i = grp; i < n; i += NGRP) {
        float mx = -__builtin_inff();
        for (int c = l; c < m; c += 8) {
            const float r = XL[i * m + c] / flen;
            mx = r > mx ? r : mx;
        }
        mx = group8_max(mx);
        auto rb_of = [&](int c) {
            const float r = XL[i * m + c] / flen;
            const float lg = is_test ? (r == mx ? 1.0f : 0.0f) : (r > 0.01f ? 1.0f : 0.0f);
            return r * lg;                                             // (:130)
        };
        float ms = -__builtin_inff();
        for (int c = l; c < m; c += 8) {
            const float r = XL[i * m + c] / flen;
            const float rb = rb_of(c);
            const float rc = r < 0.0f ? 0.0f : (r > 1.0f ? 1.0f : r);
            const float v = rc * (-PL[i * m + c]);                     // (:146)
            ms = v > ms ? v : ms;
            Rb_b[(int64_t)i * PpS + c] = rb;
            if (R_b) R_b[(int64_t)i * PpS + c] = r;
        }
        ms = group8_max(ms);
        const float ds = torder::inner_sum_group8(m, l, [&](long q) {     // (score * Rb).sum(1) (:147)
            const float sc = q < Nb ? score_p[(int64_t)b * N + q] : 0.0f;
            return sc * rb_of((int)q);
        });
        if (l == 0) {
            match_score[(int64_t)b * M + i] = ms;
            det_score[(int64_t)b * M + i] = ds;
        }
    }
}

int launch_relax_match_wide(const float *cos_in, const int32_t *inter, const int32_t *area_p, const int32_t *area_t,
                            const float *score_p, int B, int N, int M, const int32_t *n_valid, const int32_t *m_valid,
                            float w_feat, float w_iou, RelaxParams prm, int is_test, float *sim_out, float *R_out,
                            float *Rb_out, float *match_score, float *det_score, int32_t *iters_out, float *X_final,
                            float *scratch, hipStream_t stream) {
    const int PpS = N > M ? N : M + 1;
    const int64_t stride = (int64_t)wide_scratch_floats(M, PpS);
    // register-resident form while a thread's share of the table is <= 12 elements and the two LDS tables fit
    {
        const size_t cap = (size_t)M * PpS;
        const size_t lds_reg = sizeof(float) * (3 * cap + (size_t)PpS + 2 * (size_t)M + 16);
        if (cap <= 24 * (size_t)kWideRegThreads && lds_reg <= 150 * 1024 && M < 65536 && PpS < 65536) {
#define DMM_WREG(K_)                                                                                                    \
    do {                                                                                                                \
        if (lds_reg > 64 * 1024) {                                                                                      \
            hipError_t e = hipFuncSetAttribute((const void *)relax_match_wide_reg_kernel<K_>,                           \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_reg);               \
            if (e != hipSuccess) { set_last_hip_error((int)e); return DMM_ERR_LAUNCH; }                                 \
        }                                                                                                               \
        hipLaunchKernelGGL(relax_match_wide_reg_kernel<K_>, dim3(B), dim3(kWideRegThreads), lds_reg, stream, cos_in,    \
                           inter, area_p, area_t, score_p, N, M, n_valid, m_valid, w_feat, w_iou, prm, is_test, sim_out, \
                           R_out, Rb_out, match_score, det_score, iters_out, X_final, scratch, stride);                 \
    } while (0)
            if (cap <= 8 * (size_t)kWideRegThreads) DMM_WREG(8);
            else if (cap <= 16 * (size_t)kWideRegThreads) DMM_WREG(16);
            else DMM_WREG(24);
#undef DMM_WREG
            return check_launch();
        }
    }
    // the iterate X is what the column / row sums read element by element: in LDS while M x Pp floats fit (152 KB), else
    // with the rest of the state in the L2-resident scratch (300 x 40 at 20 x 5: 3.6 ms from L2, see tools/wide_timing.py)
    size_t lds = sizeof(float) * (size_t)M * PpS;
    const int x_in_lds = lds <= 152 * 1024;
    if (!x_in_lds) lds = 0;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)relax_match_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) { set_last_hip_error((int)e); return DMM_ERR_LAUNCH; }
    }
    hipLaunchKernelGGL(relax_match_wide_kernel, dim3(B), dim3(kWideSolverThreads), lds, stream, cos_in, inter, area_p,
                       area_t, score_p, N, M, n_valid, m_valid, w_feat, w_iou, prm, is_test, sim_out, R_out, Rb_out,
                       match_score, det_score, iters_out, X_final, scratch, stride, x_in_lds);
    return check_launch();
}

// ---------------------------------------------------------------------------------------------
// Backward of the solver for ANY table size: d loss / d sim from d Rb, d match_score, d det_score -- the reference's autograd
// through relax_matching (relax_match.py:36-105) and the epilogue of match_with_first_frame (match_model.py:118-147), which
// is unbounded; the register-resident kernel of dmm_solve.hip covers M <= 32, Pp <= 256.  Same scheme as that kernel: the
// forward is re-run from the saved sim (same operations in the same order as relax_match_wide_kernel -> same iterates, same
// exits), every projection sweep tapes one byte per element (the relu's gate) and per column (the column projection's
// gate), then the tape is walked backwards: given its gates a sweep is a linear map.  One workgroup per frame, all state
// in the caller's workspace (wide_bwd_bytes).  Correctness path: dependent chains through L2.
// ---------------------------------------------------------------------------------------------
constexpr int kWideBwdTables = 15;        // C X P0 P1 P2 Xs acc tmp | gX gP0 gP1 gP2 gC gl gy1

size_t wide_bwd_bytes(int N, int M, int max_iter, int proj_iter) {
    const size_t PpS = (size_t)(N > M ? N : M + 1), cap = (size_t)M * PpS;
    const size_t floats = kWideBwdTables * cap + 4 * PpS + 6 * (size_t)M + 64;
    const size_t ints = (size_t)M + (size_t)max_iter + 64;
    const size_t tape = (size_t)max_iter * (size_t)proj_iter * (cap + PpS);
    return (4 * (floats + ints) + tape + 255) / 256 * 256;
}

__device__ __forceinline__ float wide_wave_sum(float v) { return wave_sum(v); }

__global__ __launch_bounds__(kWideSolverThreads) void relax_match_bwd_wide_kernel(
    const float *__restrict__ sim_in, const float *__restrict__ score_p, int N, int M,
    const int32_t *__restrict__ n_valid, const int32_t *__restrict__ m_valid, RelaxParams prm, int is_test,
    const float *__restrict__ dRb_in, const float *__restrict__ dms_in, const float *__restrict__ dds_in,
    float *__restrict__ dsim_out, unsigned char *__restrict__ ws, size_t ws_stride) {
    __shared__ float sh[kWideSolverThreads / 64 + 1];
    const int b = blockIdx.x, tid = threadIdx.x, l = tid & 7, grp = tid >> 3, lane = tid & 63, wave = tid >> 6;
    constexpr int NT = kWideSolverThreads, NGRP = kWideSolverThreads / 8, NW = kWideSolverThreads / 64;
    const int Nb = n_valid ? n_valid[b] : N;
    const int Mb = m_valid ? m_valid[b] : M;
    const int PpS = N > M ? N : M + 1;
    float *dsim_b = dsim_out + (int64_t)b * M * N;
    if (Mb <= 0 || Nb <= 0) {
        for (int e = tid; e < M * N; e += NT) dsim_b[e] = 0.0f;
        return;
    }
    const int n = Mb, m = Nb > Mb ? Nb : Mb + 1, cnt = n * m;
    const size_t cap = (size_t)M * PpS;
    float *base = reinterpret_cast<float *>(ws + (size_t)b * ws_stride);
    float *C = base, *X = C + cap, *P0 = X + cap, *P1 = P0 + cap, *P2 = P1 + cap, *Xs = P2 + cap, *acc = Xs + cap,
          *tmp = acc + cap, *gX = tmp + cap, *gP0 = gX + cap, *gP1 = gP0 + cap, *gP2 = gP1 + cap, *gC = gP2 + cap,
          *gl = gC + cap, *gy1 = gl + cap;
    float *tc = gy1 + cap, *cg = tc + PpS, *aux_c = cg + PpS, *aux_c2 = aux_c + PpS;
    float *rt = aux_c2 + PpS, *rs = rt + M, *rmaxv = rs + M, *vmaxv = rmaxv + M, *gdir = vmaxv + M, *aux_r = gdir + M;
    int *idx = reinterpret_cast<int *>(aux_r + M + 64), *sweeps = idx + M;
    unsigned char *tape = reinterpret_cast<unsigned char *>(sweeps + prm.max_iter + 64);
    const size_t tape_stride = (size_t)cnt + (size_t)m;               // per sweep: cnt relu gates, then m column gates
    (void)aux_c; (void)aux_c2; (void)aux_r; (void)gdir;

    // ---- C = -sim_pad ----
    for (int e = tid; e < cnt; e += NT) {
        const int i = e / m, c = e - i * m;
        const float sv = c < Nb ? sim_in[(int64_t)b * M * N + (int64_t)i * N + c] : 0.0f;
        C[e] = -sv;
    }
    __syncthreads();
    // ---- greedy init (no gradient) ----
    {
        float cm = -__builtin_inff();
        for (int e = tid; e < cnt; e += NT) cm = C[e] > cm ? C[e] : cm;
        const float cmax = wide_block_max(cm, sh);
        for (int c = tid; c < m; c += NT) {
            int best = 0;
            float bv = C[c];
            for (int i = 1; i < n; ++i)
                if (C[i * m + c] < bv) { bv = C[i * m + c]; best = i; }
            for (int i = 0; i < n; ++i) tmp[i * m + c] = i == best ? C[i * m + c] : cmax;
        }
        __syncthreads();
        for (int i = tid; i < n; i += NT) {
            int best = 0;
            float bv = tmp[i * m];
            for (int c = 1; c < m; ++c)
                if (tmp[i * m + c] < bv) { bv = tmp[i * m + c]; best = c; }
            idx[i] = best;
        }
        __syncthreads();
        for (int e = tid; e < cnt; e += NT) {
            const int i = e / m, c = e - i * m;
            const float x0 = c == idx[i] ? 1.0f : 0.0f;
            X[e] = x0;
            acc[e] = 0.0f + x0;
            P0[e] = 0.0f; P1[e] = 0.0f; P2[e] = 0.0f;
        }
    }
    __syncthreads();
    // ---- forward, taped (the loop of relax_match_wide_kernel) ----
    const float fn = (float)n, fm = (float)m;
    const int cbound = torder::outer_class_bound(m);
    int len = 1, pos = 0;
    float cost_prev = 0.0f;
    for (int it = 0; it < prm.max_iter; ++it) {
        for (int e = tid; e < cnt; e += NT) {
            const float g = prm.lr * C[e];
            const float x = X[e] - g;
            X[e] = x;
            tmp[e] = x * C[e];
            acc[e] = acc[e] + x;
        }
        __syncthreads();
        if (tid < 8) {
            const float c = torder::norm2_group8(cnt, tid, [&](long i) { return tmp[i]; });
            if (tid == 0) sh[NW] = c;
        }
        __syncthreads();
        const float cost = sh[NW];
        ++len;
        int ns = 0;
        for (int j = 0; j < prm.proj_iter; ++j) {
            unsigned char *tp = tape + (size_t)pos * tape_stride;
            for (int e = tid; e < cnt; e += NT) {
                const float xs = X[e];
                Xs[e] = xs;
                const float x = xs + P0[e];
                const float y = x > 0.0f ? x : 0.0f;
                tp[e] = x > 0.0f ? 1 : 0;                              // relu gate
                P0[e] = x - y;
                X[e] = y + P1[e];
            }
            __syncthreads();
            for (int c = tid; c < m; c += NT) {
                const float cs = torder::outer_sum_col(n, c < cbound, [&](long i) { return X[i * m + c]; });
                const bool over = !(cs <= 1.0f);
                tp[cnt + c] = over ? 1 : 0;                            // column gate
                tc[c] = cs <= 1.0f ? 0.0f : (cs - 1.0f) / fn;
            }
            __syncthreads();
            for (int e = tid; e < cnt; e += NT) {
                const int c = e % m;
                const float x = X[e];
                const float y = x - tc[c];
                P1[e] = x - y;
                X[e] = y + P2[e];
            }
            __syncthreads();
            for (int i = grp; i < n; i += NGRP) {
                const float s = torder::inner_sum_group8(m, l, [&](long k) { return X[i * m + k]; });
                if (l == 0) rt[i] = (s - 1.0f) / fm;
            }
            __syncthreads();
            int moved = 0;
            for (int e = tid; e < cnt; e += NT) {
                const int i = e / m;
                const float x = X[e];
                const float y = x - rt[i];
                P2[e] = x - y;
                X[e] = y;
                const float d = y - Xs[e];
                const float sq = d * d;
                moved |= !(sq == 0.0f);
            }
            ++pos;
            ++ns;
            if (!__syncthreads_or(moved)) break;
        }
        if (tid == 0) sweeps[it] = ns;
        if (cost_prev == cost) break;
        cost_prev = cost;
    }
    const int iters = len - 1;
    __syncthreads();

    // ---- epilogue adjoints: R = acc / len; logic; Rb; match_score = max_c clamp(R) * sim_pad; det_score = sum_c sc * Rb ----
    const float flen = (float)len;
    for (int i = wave; i < n; i += NW) {                               // row maxima of R and of clamp(R) * sim_pad
        float mx = -__builtin_inff(), vx = -__builtin_inff();
        for (int c = lane; c < m; c += 64) {
            const float r = acc[i * m + c] / flen;
            const float rc = r < 0.0f ? 0.0f : (r > 1.0f ? 1.0f : r);
            const float v = rc * (-C[i * m + c]);
            mx = r > mx ? r : mx;
            vx = v > vx ? v : vx;
        }
        mx = wave_max(mx);
        vx = wave_max(vx);
        int cand = 0x7fffffff;                                         // torch.max(dim) backward: the first maximal index
        for (int c = lane; c < m; c += 64) {
            const float r = acc[i * m + c] / flen;
            const float rc = r < 0.0f ? 0.0f : (r > 1.0f ? 1.0f : r);
            if (rc * (-C[i * m + c]) == vx && c < cand) cand = c;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const int o = __shfl_xor(cand, d);
            cand = o < cand ? o : cand;
        }
        if (lane == 0) { rmaxv[i] = mx; idx[i] = cand; }
    }
    __syncthreads();
    for (int e = tid; e < cnt; e += NT) {
        const int i = e / m, c = e - i * m;
        const float r = acc[e] / flen;
        const float lg = is_test ? (r == rmaxv[i] ? 1.0f : 0.0f) : (r > 0.01f ? 1.0f : 0.0f);
        const float sc = c < Nb ? score_p[(int64_t)b * N + c] : 0.0f;
        const float dms = dms_in ? dms_in[(int64_t)b * M + i] : 0.0f;
        const float dds = dds_in ? dds_in[(int64_t)b * M + i] : 0.0f;
        float dR = ((dRb_in ? dRb_in[(int64_t)b * M * PpS + (int64_t)i * PpS + c] : 0.0f) + dds * sc) * lg;
        float gd = 0.0f;
        if (c == idx[i]) {
            const float rc = r < 0.0f ? 0.0f : (r > 1.0f ? 1.0f : r);
            if (r >= 0.0f && r <= 1.0f) dR += dms * (-C[e]);            // clamp passes its gradient on [0, 1]
            gd = dms * rc;                                             // d / d sim_pad of clamp(R) * sim_pad
        }
        gl[e] = dR / flen;
        tmp[e] = gd;                                                   // (tmp is free now: the direct term)
        gX[e] = 0.0f; gP0[e] = 0.0f; gP1[e] = 0.0f; gP2[e] = 0.0f; gC[e] = 0.0f;
    }
    __syncthreads();

    // ---- reverse sweep through the tape ----
    const float inv_m = 1.0f / fm, inv_n = 1.0f / fn;
    for (int it = iters - 1; it >= 0; --it) {
        const int ns = sweeps[it];
        for (int sidx = 0; sidx < ns; ++sidx) {
            --pos;
            const unsigned char *tp = tape + (size_t)pos * tape_stride;
            // row projection: y2 = c - (rowsum(c) - 1) / m; P2' = c - y2; X' = y2
            for (int i = wave; i < n; i += NW) {
                float s = 0.0f;
                for (int c = lane; c < m; c += 64) s += gX[i * m + c] - gP2[i * m + c];
                s = wide_wave_sum(s);
                if (lane == 0) rs[i] = s;
            }
            __syncthreads();
            for (int e = tid; e < cnt; e += NT) {
                const int i = e / m;
                const float gy2 = gX[e] - gP2[e];
                const float gc = gP2[e] + gy2 - rs[i] * inv_m;
                gP2[e] = gc;                                           // c = y1 + P2
                gy1[e] = gc - gP1[e];                                  // P1' = b - y1
            }
            __syncthreads();
            // column projection: y1 = b - over * (colsum(b) - 1) / n
            for (int c = tid; c < m; c += NT) {
                float s = 0.0f;
                for (int i = 0; i < n; ++i) s += gy1[i * m + c];
                cg[c] = tp[cnt + c] ? s * inv_n : 0.0f;
            }
            __syncthreads();
            for (int e = tid; e < cnt; e += NT) {
                const int c = e % m;
                const float gb = gP1[e] + gy1[e] - cg[c];
                gP1[e] = gb;                                           // b = y0 + P1
                const float gy0 = gb - gP0[e];                         // P0' = a - y0
                const float ga = gP0[e] + (tp[e] ? gy0 : 0.0f);        // y0 = relu(a)
                gX[e] = ga;                                            // a = X + P0
                gP0[e] = ga;
            }
            __syncthreads();
        }
        // gradient step X_pre = X_prev - lr * C; X_pre is this iteration's entry of X_list
        for (int e = tid; e < cnt; e += NT) {
            const float g = gX[e] + gl[e];
            gC[e] = gC[e] - prm.lr * g;
            gX[e] = g;
        }
        __syncthreads();
    }
    // C = -sim_pad -> dsim = -dC (+ the direct match_score term); padded columns are dropped, dead entries zero
    for (int e = tid; e < M * N; e += NT) {
        const int i = e / N, c = e - i * N;
        dsim_b[e] = (i < n && c < Nb) ? tmp[i * m + c] - gC[i * m + c] : 0.0f;
    }
}

int launch_relax_match_bwd_wide(const float *sim, const float *score_p, int B, int N, int M, const int32_t *n_valid,
                                const int32_t *m_valid, RelaxParams prm, int is_test, const float *dRb, const float *dms,
                                const float *dds, float *dsim_out, void *workspace, hipStream_t stream) {
    const size_t stride = wide_bwd_bytes(N, M, prm.max_iter, prm.proj_iter);
    hipLaunchKernelGGL(relax_match_bwd_wide_kernel, dim3(B), dim3(kWideSolverThreads), 0, stream, sim, score_p, N, M, n_valid,
                       m_valid, prm, is_test, dRb, dms, dds, dsim_out, (unsigned char *)workspace, stride);
    return check_launch();
}

}  // namespace dmm
