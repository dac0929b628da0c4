/*
 * oracle/dmm_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the DMM-Net matching layer (ZENGXH/DMM_Net).  It is the
 * checker the HIP path is compared against; nothing in the product path
 * (dmm_net_amd/) may link, import or call it.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it.
 *
 * Parity status: PINNED.  Every function below is checked against golden vectors
 * captured by importing the reference's own Python modules in the build container
 * (tests/golden/gen_golden.py -> the .npz fixtures in tests/golden; tests/test_oracle_golden.py).
 *
 * Each function cites the reference file:line it restates (paths relative to the
 * reference root).  Arithmetic notes:
 *   - elementwise fp32 ops are done one IEEE op at a time (build with
 *     -ffp-contract=off so the compiler cannot fuse a*b+c);
 *   - row / column sums and norms follow the summation ORDER of the torch CPU kernels the
 *     golden vectors were captured with (torch 2.10 CPU, AVX2 dispatch, 8-lane vectors:
 *     cascade sum of ATen SumKernel, 2-norm fast path of ReduceOpsKernel; restated below
 *     from their published algorithm, see t_* helpers).  With that order the solver, the
 *     cosine and every [M,N]-sized table are BIT-EXACT against the goldens -- which matters
 *     because relax_matching's early exits compare fp32 norms for exact equality
 *     (relax_match.py:88,96) and are chaotic in the last ulp once the iteration has
 *     converged (the reference's own self-test exits at step 57 only under this order);
 *   - IoU counts are integers, and iou = inter / (union + 1e-6f) is bit exact.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define DMMO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------
 * compute_iou_binary_mask_2D  (dmm/utils/match_helper.py:9-28) as called by
 * MatchModel.compute_cost_matrix (dmm/modules/match_model.py:83-89): every proposal
 * plane against every template plane.  a = x > 0.5 (strict), inter = sum(a & b),
 * union = sum(a | b).  Returned as integer tables: inter[M][N], area_p[N], area_t[M]
 * (union = area_p + area_t - inter).
 * ---------------------------------------------------------------------------------- */
DMMO_API void dmmo_iou_counts(const float *P, int N, const float *T, int M, int HW,
                              int32_t *inter, int32_t *area_p, int32_t *area_t) {
    uint8_t *bp = (uint8_t *)malloc((size_t)N * HW + 1);
    uint8_t *bt = (uint8_t *)malloc((size_t)M * HW + 1);
    for (int n = 0; n < N; ++n) {
        int32_t a = 0;
        for (int x = 0; x < HW; ++x) {
            uint8_t b = P[(size_t)n * HW + x] > 0.5f;
            bp[(size_t)n * HW + x] = b;
            a += b;
        }
        area_p[n] = a;
    }
    for (int m = 0; m < M; ++m) {
        int32_t a = 0;
        for (int x = 0; x < HW; ++x) {
            uint8_t b = T[(size_t)m * HW + x] > 0.5f;
            bt[(size_t)m * HW + x] = b;
            a += b;
        }
        area_t[m] = a;
    }
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            const uint8_t *a = bt + (size_t)m * HW, *b = bp + (size_t)n * HW;
            int32_t s = 0;
            for (int x = 0; x < HW; ++x) s += a[x] & b[x];
            inter[m * N + n] = s;
        }
    free(bp);
    free(bt);
}

/* match_helper.py:22-27: union = union.sum(1) + 1e-6 (fp32 add); iou = inter / union. */
DMMO_API void dmmo_iou_from_counts(const int32_t *inter, const int32_t *area_p,
                                   const int32_t *area_t, int N, int M, float *iou) {
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            int32_t i = inter[m * N + n];
            float u = (float)(area_p[n] + area_t[m] - i) + 1e-6f;
            iou[m * N + n] = (float)i / u;
        }
}

/* ====================================================================================
 * Reduction orders of the torch CPU kernels (third-party arithmetic: PyTorch 2.10,
 * aten/src/ATen/native/cpu/SumKernel.cpp "cascade_sum" and ReduceOpsKernel.cpp
 * "norm_kernel_tensor_iterator_impl", AVX2 dispatch => 8 fp32 lanes).  Restated from the
 * published algorithm; validated bit-for-bit through the golden vectors (every fp32 table of
 * every fixture, tests/test_oracle_golden.py).
 * ================================================================================== */
#define TV 8 /* Vectorized<float>::size() under the AVX2 dispatch */

static int ceil_log2(long x) {
    int l = 0;
    while ((1L << l) < x) ++l;
    return x <= 1 ? 0 : l;
}

/* multi_row_sum: cascade-sum `size` rows into nacc = nchunk*w accumulators.
 * element (row i, chunk k, lane l) = in[i*row_stride + k*chunk_stride + l]. */
static void t_multi_row_sum(const float *in, long row_stride, int nchunk, long chunk_stride, int w,
                            long size, float *out) {
    enum { LEVELS = 4, MAXACC = 32 };
    float acc[LEVELS][MAXACC];
    const int nacc = nchunk * w;
    int lp = ceil_log2(size) / LEVELS;
    const int level_power = lp > 4 ? lp : 4;
    const long level_step = 1L << level_power, level_mask = level_step - 1;
    for (int j = 0; j < LEVELS; ++j)
        for (int a = 0; a < nacc; ++a) acc[j][a] = 0.0f;
    long i = 0;
    for (; i + level_step <= size;) {
        for (long j = 0; j < level_step; ++j, ++i)
            for (int k = 0; k < nchunk; ++k)
                for (int l = 0; l < w; ++l)
                    acc[0][k * w + l] += in[i * row_stride + k * chunk_stride + l];
        for (int j = 1; j < LEVELS; ++j) {
            for (int a = 0; a < nacc; ++a) {
                acc[j][a] += acc[j - 1][a];
                acc[j - 1][a] = 0.0f;
            }
            const long mask = level_mask << (j * level_power);
            if ((i & mask) != 0) break;
        }
    }
    for (; i < size; ++i)
        for (int k = 0; k < nchunk; ++k)
            for (int l = 0; l < w; ++l) acc[0][k * w + l] += in[i * row_stride + k * chunk_stride + l];
    for (int j = 1; j < LEVELS; ++j)
        for (int a = 0; a < nacc; ++a) acc[0][a] += acc[j][a];
    for (int a = 0; a < nacc; ++a) out[a] = acc[0][a];
}

/* row_sum: `size` items of width w (item i at in + i*item_stride), 4-way ILP split. */
static void t_row_sum(const float *in, long item_stride, int w, long size, float *out) {
    float ps[4 * TV];
    const long size_ilp = size / 4;
    t_multi_row_sum(in, item_stride * 4, 4, item_stride, w, size_ilp, ps);
    for (long i = size_ilp * 4; i < size; ++i)
        for (int l = 0; l < w; ++l) ps[l] += in[i * item_stride + l];
    for (int k = 1; k < 4; ++k)
        for (int l = 0; l < w; ++l) ps[l] += ps[k * w + l];
    for (int l = 0; l < w; ++l) out[l] = ps[l];
}

/* sum over a contiguous run of n floats (reduced dim is the fastest one). */
static float t_inner_sum(const float *x, long n) {
    float r;
    if (n < TV) { /* scalar_inner_sum */
        t_row_sum(x, 1, 1, n, &r);
        return r;
    }
    float vec[TV];
    const long vs = n / TV;
    t_row_sum(x, TV, TV, vs, vec);
    float acc = 0.0f;
    for (long k = vs * TV; k < n; ++k) acc += x[k];
    for (int k = 0; k < TV; ++k) acc += vec[k];
    return acc;
}

/* sum over the rows of X[n][m] (row stride `rs`), one result per column. */
static void t_outer_sum(const float *X, long n, long m, long rs, float *out) {
    long j = 0;
    if (m >= TV) { /* vectorized_outer_sum */
        for (; j + 4 * TV <= m; j += 4 * TV) t_multi_row_sum(X + j, rs, 1, 0, 4 * TV, n, out + j);
        for (; j + TV <= m; j += TV) t_row_sum(X + j, rs, TV, n, out + j);
        for (; j < m; ++j) t_row_sum(X + j, rs, 1, n, out + j);
    } else { /* scalar_outer_sum */
        for (; j + 3 < m; j += 4) t_multi_row_sum(X + j, rs, 1, 0, 4, n, out + j);
        for (; j < m; ++j) t_row_sum(X + j, rs, 1, n, out + j);
    }
}

/* 2-norm of a contiguous run: 8 fma lanes, lanes added in order, tail in groups of 4
 * (square rounded, then added) and a final <4 remainder with fma. */
static float t_norm2(const float *x, long n) {
    float acc[TV];
    for (int l = 0; l < TV; ++l) acc[l] = 0.0f;
    long d = 0;
    for (; d < n - (n % TV); d += TV)
        for (int l = 0; l < TV; ++l) acc[l] = fmaf(x[d + l], x[d + l], acc[l]);
    float b = acc[0];
    for (int l = 1; l < TV; ++l) b = b + acc[l];
    for (; n - d >= 4; d += 4)
        for (int l = 0; l < 4; ++l) {
            float sq = x[d + l] * x[d + l];
            b = b + sq;
        }
    for (; d < n; ++d) b = fmaf(x[d], x[d], b);
    return sqrtf(b);
}

/* ------------------------------------------------------------------------------------
 * get_cosine_score (dmm/utils/match_helper.py:51-64): F.cosine_similarity over D of the
 * expanded [O,D,P] tensors.  Semantics of the torch in the build container (2.10):
 * each vector is divided by max(|v|, eps) (eps = 1e-8) first, then the products are summed
 * over D:  cos[o,p] = sum_d (q[o,d]/qn[o]) * (k[p,d]/kn[p]).
 * q = template features [M,D], k = proposal features [N,D]; out [M][N].
 * ---------------------------------------------------------------------------------- */
DMMO_API void dmmo_cosine(const float *q, const float *k, int M, int N, int D, float *out) {
    const float eps = 1e-8f;
    float *qq = (float *)malloc(sizeof(float) * ((size_t)(M + N) * D + (size_t)D * N));
    float *kk = qq + (size_t)M * D, *prod = kk + (size_t)N * D;
    for (int m = 0; m < M; ++m) {
        float nr = t_norm2(q + (size_t)m * D, D);
        nr = nr > eps ? nr : eps;
        for (int d = 0; d < D; ++d) qq[(size_t)m * D + d] = q[(size_t)m * D + d] / nr;
    }
    for (int n = 0; n < N; ++n) {
        float nr = t_norm2(k + (size_t)n * D, D);
        nr = nr > eps ? nr : eps;
        for (int d = 0; d < D; ++d) kk[(size_t)n * D + d] = k[(size_t)n * D + d] / nr;
    }
    for (int m = 0; m < M; ++m) {
        /* prod[d][n] is the contiguous [D,P] slab torch reduces over d for this o */
        for (int d = 0; d < D; ++d)
            for (int n = 0; n < N; ++n)
                prod[(size_t)d * N + n] = qq[(size_t)m * D + d] * kk[(size_t)n * D + d];
        if (N == 1)
            out[m] = t_inner_sum(prod, D); /* [O,D,1]: the reduced dim is the fastest one */
        else
            t_outer_sum(prod, D, N, N, out + (size_t)m * N);
    }
    free(qq);
}

/* ------------------------------------------------------------------------------------
 * Greedy "row min" initialisation of relax_matching (relax_match.py:45-55).
 * C is [n rows = templates][m cols = proposals].  For each column keep only its FIRST
 * minimal row, every other entry becomes max(C); then each row picks its FIRST minimal
 * column.  A row owning no column minimum is all-max and picks column 0.
 * ---------------------------------------------------------------------------------- */
DMMO_API void dmmo_greedy_init(const float *C, int n, int m, int32_t *idx) {
    float cmax = C[0];
    for (int i = 0; i < n * m; ++i)
        if (C[i] > cmax) cmax = C[i];
    float *crm = (float *)malloc(sizeof(float) * n * m);
    for (int j = 0; j < m; ++j) {
        int best = 0;
        for (int i = 1; i < n; ++i)
            if (C[i * m + j] < C[best * m + j]) best = i; /* first argmin */
        for (int i = 0; i < n; ++i) crm[i * m + j] = (i == best) ? C[i * m + j] : cmax;
    }
    for (int i = 0; i < n; ++i) {
        int best = 0;
        for (int j = 1; j < m; ++j)
            if (crm[i * m + j] < crm[i * m + best]) best = j;
        idx[i] = best;
    }
    free(crm);
}

/* ------------------------------------------------------------------------------------
 * relax_matching (relax_match.py:36-105) with project_row (:9-19) / project_col (:21-34).
 * Projected gradient descent + Dykstra cyclic projections onto
 * {X>=0} n {col sums <= 1} n {row sums = 1}.  The Dykstra increments P0..P2 persist
 * across outer iterations (:62).  X_list holds the PRE-projection iterates (:71).
 *
 * Outputs (any may be NULL): X_out final projected X [n*m]; R_mean_out =
 * sum(X_list)/len(X_list) as MatchModel does (match_model.py:121) [n*m];
 * cost_out [max_iter+1] (cost[0] = 0); xlist_out [(max_iter+1)*n*m];
 * inner_sweeps_out [max_iter] sweeps executed per outer iteration.
 * Returns the number of executed outer iterations (= len(X_list) - 1).
 * ---------------------------------------------------------------------------------- */
DMMO_API int dmmo_relax(const float *C, int n, int m, int max_iter, int proj_iter, float lr,
                        float *X_out, float *R_mean_out, float *cost_out, float *xlist_out,
                        int32_t *inner_sweeps_out) {
    const int cnt = n * m;
    float *buf = (float *)calloc((size_t)cnt * 8 + m, sizeof(float));
    float *X = buf, *Y = buf + cnt, *P0 = buf + 2 * cnt, *P1 = buf + 3 * cnt, *P2 = buf + 4 * cnt,
          *Xs = buf + 5 * cnt, *acc = buf + 6 * cnt, *tmp = buf + 7 * cnt, *cs = buf + 8 * cnt;
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * n);
    dmmo_greedy_init(C, n, m, idx);
    for (int i = 0; i < n; ++i) X[i * m + idx[i]] = 1.0f;
    free(idx);

    /* python: sum(X_list) = ((0 + X0) + X1) + ... sequential fp32 adds */
    for (int i = 0; i < cnt; ++i) acc[i] = 0.0f + X[i];
    if (xlist_out) memcpy(xlist_out, X, sizeof(float) * cnt);
    int len = 1;
    float cost_prev = 0.0f;
    if (cost_out) cost_out[0] = 0.0f;

    for (int it = 0; it < max_iter; ++it) {
        for (int i = 0; i < cnt; ++i) {
            float g = lr * C[i];
            X[i] = X[i] - g;                                        /* :69 */
        }
        for (int i = 0; i < cnt; ++i) tmp[i] = X[i] * C[i];
        float cost = t_norm2(tmp, cnt);                             /* :70 */
        if (cost_out) cost_out[it + 1] = cost;
        for (int i = 0; i < cnt; ++i) acc[i] = acc[i] + X[i];       /* :71 */
        if (xlist_out) memcpy(xlist_out + (size_t)len * cnt, X, sizeof(float) * cnt);
        ++len;
        int sweeps = 0;
        for (int j = 0; j < proj_iter; ++j) {
            memcpy(Xs, X, sizeof(float) * cnt);                     /* :73 */
            /* relu set (:74-76) */
            for (int i = 0; i < cnt; ++i) {
                float x = X[i] + P0[i];
                float y = x > 0.0f ? x : 0.0f;
                P0[i] = x - y;
                Y[i] = y;
            }
            /* column set (:78-80, project_col :21-34) */
            for (int i = 0; i < cnt; ++i) X[i] = Y[i] + P1[i];
            t_outer_sum(X, n, m, m, cs);
            for (int c = 0; c < m; ++c) {
                if (cs[c] <= 1.0f) {
                    for (int r = 0; r < n; ++r) Y[r * m + c] = X[r * m + c];
                } else {
                    float t = (cs[c] - 1.0f) / (float)n;
                    for (int r = 0; r < n; ++r) Y[r * m + c] = X[r * m + c] - t;
                }
            }
            for (int i = 0; i < cnt; ++i) P1[i] = X[i] - Y[i];
            /* row set (:82-84, project_row :9-19) */
            for (int i = 0; i < cnt; ++i) X[i] = Y[i] + P2[i];
            for (int r = 0; r < n; ++r) {
                float s = t_inner_sum(X + (size_t)r * m, m);
                float t = (s - 1.0f) / (float)m;
                for (int c = 0; c < m; ++c) Y[r * m + c] = X[r * m + c] - t;
            }
            for (int i = 0; i < cnt; ++i) P2[i] = X[i] - Y[i];
            memcpy(X, Y, sizeof(float) * cnt);                      /* :86 */
            ++sweeps;
            for (int i = 0; i < cnt; ++i) tmp[i] = X[i] - Xs[i];
            if (t_norm2(tmp, cnt) == 0.0f) break;                   /* :88-89 */
        }
        if (inner_sweeps_out) inner_sweeps_out[it] = sweeps;
        if (cost_prev == cost) break;                               /* :96-98 */
        cost_prev = cost;
    }
    if (X_out) memcpy(X_out, X, sizeof(float) * cnt);
    if (R_mean_out)
        for (int i = 0; i < cnt; ++i) R_mean_out[i] = acc[i] / (float)len;
    free(buf);
    return len - 1;
}

/* ------------------------------------------------------------------------------------
 * compute_matching_loss (match_helper.py:30-49), training only:
 * gt_iou[O,P] = IoU(proposal>0.5, targets>0.5); gt_matched = greedy init of -gt_iou
 * (relax_matching(..., 0,0,0)[0]); loss = F.mse_loss(feature_sim, gt_matched) =
 * mean((feature_sim - gt_matched)^2).  gt_iou_out / gt_onehot_out [M*N] optional.
 * ---------------------------------------------------------------------------------- */
DMMO_API float dmmo_matching_loss(const float *P, int N, const float *targets, int M, int HW,
                                  const float *feature_sim, float *gt_iou_out,
                                  float *gt_onehot_out) {
    int32_t *inter = (int32_t *)malloc(sizeof(int32_t) * (M * N + N + M));
    int32_t *ap = inter + M * N, *at = ap + N;
    float *iou = (float *)malloc(sizeof(float) * M * N * 3);
    float *negc = iou + M * N, *sq = negc + M * N;
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * M);
    dmmo_iou_counts(P, N, targets, M, HW, inter, ap, at);
    dmmo_iou_from_counts(inter, ap, at, N, M, iou);
    for (int i = 0; i < M * N; ++i) negc[i] = -iou[i];
    dmmo_greedy_init(negc, M, N, idx);
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            float g = (idx[m] == n) ? 1.0f : 0.0f;
            if (gt_onehot_out) gt_onehot_out[m * N + n] = g;
            float d = feature_sim[m * N + n] - g;
            sq[m * N + n] = d * d;
        }
    float loss = t_inner_sum(sq, (long)M * N) / (float)(M * N);
    if (gt_iou_out) memcpy(gt_iou_out, iou, sizeof(float) * M * N);
    free(inter);
    free(iou);
    free(idx);
    return loss;
}

/* ------------------------------------------------------------------------------------
 * MatchModel.forward for one frame (match_model.py:24-148), algo 'relax'.
 *   sim = (1-w)*cos + w*iou                               (:90)
 *   pad to [O, O+1] with zeros if P <= O                  (:109-113)
 *   R = mean(X_list) of relax_matching(-sim_pad)          (:116-121)
 *   logic = (R == rowmax) if is_test else (R > 0.01)      (:124-129);  Rb = R*logic (:130)
 *   full_outmask = Rb @ mask2d (padded rows are zero)     (:134-144)
 *   match_score = max_p clamp(R,0,1)*sim_pad              (:146)
 *   det_score   = sum_p score_p * Rb                      (:147)
 * Pp = max(P, O+1).  Output tables are [O*Pp] (sim_out / cos_out are [O*P]).
 * Returns executed outer iterations.  full_outmask accumulates in double (torch.mm's
 * blocked sgemm order is not restated; with one non-zero per row, i.e. is_test, the
 * product is exact either way).
 * ---------------------------------------------------------------------------------- */
DMMO_API int dmmo_match_forward(const float *prop_mask, const float *tplt_mask,
                                const float *prop_feat, const float *tplt_feat,
                                const float *prop_score, int P, int O, int HW, int D,
                                float score_weight, int max_iter, int proj_iter, float lr,
                                int is_test,
                                float *full_outmask, float *match_score, float *det_score,
                                float *sim_out, float *R_out, float *logic_out, float *Rb_out,
                                float *cos_out, int32_t *inter_out, int32_t *area_p_out,
                                int32_t *area_t_out) {
    const int Pp = P > O ? P : O + 1;
    int32_t *inter = (int32_t *)malloc(sizeof(int32_t) * (O * P + P + O));
    int32_t *ap = inter + O * P, *at = ap + P;
    float *cosv = (float *)malloc(sizeof(float) * (size_t)O * P * 3);
    float *iou = cosv + O * P, *sim = iou + O * P;
    float *Cm = (float *)calloc((size_t)O * Pp * 4 + Pp, sizeof(float));
    float *simp = Cm + O * Pp, *R = simp + O * Pp, *Rb = R + O * Pp, *row = Rb + O * Pp;

    dmmo_cosine(tplt_feat, prop_feat, O, P, D, cosv);
    dmmo_iou_counts(prop_mask, P, tplt_mask, O, HW, inter, ap, at);
    dmmo_iou_from_counts(inter, ap, at, P, O, iou);
    const float w1 = (float)(1.0 - (double)score_weight), w2 = score_weight;
    for (int i = 0; i < O * P; ++i) {
        float a = cosv[i] * w1, b = iou[i] * w2;
        sim[i] = a + b;
    }
    for (int o = 0; o < O; ++o)
        for (int p = 0; p < Pp; ++p) {
            float s = p < P ? sim[o * P + p] : 0.0f;
            simp[o * Pp + p] = s;
            Cm[o * Pp + p] = -s;
        }
    int iters = dmmo_relax(Cm, O, Pp, max_iter, proj_iter, lr, NULL, R, NULL, NULL, NULL);

    for (int o = 0; o < O; ++o) {
        float mx = R[o * Pp];
        for (int p = 1; p < Pp; ++p)
            if (R[o * Pp + p] > mx) mx = R[o * Pp + p];
        float ms = 0.0f;
        for (int p = 0; p < Pp; ++p) {
            float r = R[o * Pp + p];
            float lg = is_test ? (r == mx ? 1.0f : 0.0f) : (r > 0.01f ? 1.0f : 0.0f);
            float rb = r * lg;
            Rb[o * Pp + p] = rb;
            if (logic_out) logic_out[o * Pp + p] = lg;
            float rc = r < 0.0f ? 0.0f : (r > 1.0f ? 1.0f : r);
            float v = rc * simp[o * Pp + p];
            if (p == 0 || v > ms) ms = v;
            float sc = p < P ? prop_score[p] : 0.0f;
            row[p] = sc * rb;
        }
        match_score[o] = ms;
        det_score[o] = t_inner_sum(row, Pp);
    }
    if (full_outmask)
        for (int o = 0; o < O; ++o)
            for (int x = 0; x < HW; ++x) {
                double s = 0;
                for (int p = 0; p < P; ++p) {
                    float rb = Rb[o * Pp + p];
                    if (rb != 0.0f) s += (double)(rb * prop_mask[(size_t)p * HW + x]);
                }
                full_outmask[(size_t)o * HW + x] = (float)s;
            }
    if (sim_out) memcpy(sim_out, sim, sizeof(float) * O * P);
    if (cos_out) memcpy(cos_out, cosv, sizeof(float) * O * P);
    if (R_out) memcpy(R_out, R, sizeof(float) * O * Pp);
    if (Rb_out) memcpy(Rb_out, Rb, sizeof(float) * O * Pp);
    if (inter_out) memcpy(inter_out, inter, sizeof(int32_t) * O * P);
    if (area_p_out) memcpy(area_p_out, ap, sizeof(int32_t) * P);
    if (area_t_out) memcpy(area_t_out, at, sizeof(int32_t) * O);
    free(inter);
    free(cosv);
    free(Cm);
    return iters;
}

/* ------------------------------------------------------------------------------------
 * Host check of the device's division-by-constant (dmm_net_amd/csrc/dmm_common.h div_by_const):
 * q0 = a*rcp, r = fma(-q0, b, a), q = fma(r, rcp, q0) with rcp = RN(1/b) must equal the IEEE a / b the
 * reference computes (`/ X.shape[k]`, relax_match.py:19,32).  Returns the number of mismatches over
 * `samples` pseudo-random operands per divisor b = 1..bmax (exponents 2^-27 .. 2^4, both signs).
 * ---------------------------------------------------------------------------------- */
DMMO_API long dmmo_check_div_by_const(int bmax, long samples) {
    long bad = 0;
    uint64_t s = 88172645463325252ULL;
    for (int b = 1; b <= bmax; ++b) {
        const float fb = (float)b, y = 1.0f / fb;
        for (long t = 0; t < samples; ++t) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            const uint32_t u = (uint32_t)s;
            const uint32_t expo = 100 + (u >> 8) % 32;
            const uint32_t bits = (u & 0x80000000u) | (expo << 23) | ((uint32_t)(s >> 32) & 0x7fffff);
            float a;
            memcpy(&a, &bits, 4);
            const float ref = a / fb;
            const float q0 = a * y;
            const float r = fmaf(-q0, fb, a);
            const float q1 = fmaf(r, y, q0);
            bad += (q1 != ref);
        }
    }
    return bad;
}

/* ------------------------------------------------------------------------------------
 * ROI feature extractor (dmm/modules/feature_extractor.py:11-52): maskrcnn_benchmark's legacy ROIAlign
 * (third-party, github.com/ZENGXH/maskrcnn-benchmark un-pinned HEAD; algorithm restated from its published
 * ROIAlign_cuda.cu / ROIAlign_cpu.cpp: roi size clamped to >= 1, sampling_ratio x sampling_ratio bilinear
 * samples per bin averaged, samples outside [-1, size] contribute 0) on 4 levels for every roi, then the
 * mean over the pooled H x W bins.  PARITY UN-PINNED: the package is absent and the reference holds no
 * fixtures for it; this literal restatement is the checker for the fused HIP kernel.
 * feat[l]: [B, C, H[l], W[l]] fp32; rois [R,5]; out [R, 4*C].
 * ---------------------------------------------------------------------------------- */
static float bilinear_legacy(const float *data, int height, int width, float y, float x) {
    if (y < -1.0f || y > (float)height || x < -1.0f || x > (float)width) return 0.0f;
    if (y <= 0.0f) y = 0.0f;
    if (x <= 0.0f) x = 0.0f;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; } else y_high = y_low + 1;
    if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; } else x_high = x_low + 1;
    const float ly = y - (float)y_low, lx = x - (float)x_low, hy = 1.0f - ly, hx = 1.0f - lx;
    const float v1 = data[y_low * width + x_low], v2 = data[y_low * width + x_high];
    const float v3 = data[y_high * width + x_low], v4 = data[y_high * width + x_high];
    const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
    return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}

DMMO_API void dmmo_roialign4_mean(const float *const feat[4], int B, int C, const int H[4], const int W[4],
                                  const float scale[4], const float *rois, int R, int pooled, int sampling,
                                  float *out) {
    (void)B;
    for (int r = 0; r < R; ++r) {
        const float *roi = rois + 5 * r;
        const int b = (int)roi[0];
        for (int l = 0; l < 4; ++l) {
            const float sw = roi[1] * scale[l], sh = roi[2] * scale[l], ew = roi[3] * scale[l], eh = roi[4] * scale[l];
            const float rw = fmaxf(ew - sw, 1.0f), rh = fmaxf(eh - sh, 1.0f);
            const float bh = rh / (float)pooled, bw = rw / (float)pooled;
            const int gh = sampling > 0 ? sampling : (int)ceilf(rh / pooled);
            const int gw = sampling > 0 ? sampling : (int)ceilf(rw / pooled);
            const float count = (float)(gh * gw);
            for (int c = 0; c < C; ++c) {
                const float *data = feat[l] + ((size_t)b * C + c) * H[l] * W[l];
                double mean_h = 0;                                  /* .mean(4) then .mean(3) */
                for (int ph = 0; ph < pooled; ++ph) {
                    double mean_w = 0;
                    for (int pw = 0; pw < pooled; ++pw) {
                        float v = 0.0f;
                        for (int iy = 0; iy < gh; ++iy) {
                            const float y = sh + ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
                            for (int ix = 0; ix < gw; ++ix) {
                                const float x = sw + pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
                                v += bilinear_legacy(data, H[l], W[l], y, x);
                            }
                        }
                        mean_w += (double)(v / count);
                    }
                    mean_h += mean_w / pooled;
                }
                out[(size_t)r * 4 * C + (size_t)l * C + c] = (float)(mean_h / pooled);
            }
        }
    }
}

/* ------------------------------------------------------------------------------------
 * Proposal preprocessing.
 *
 * paste_mask_in_image + binmask_to_box (dmm/utils/masker.py:110-173): the M x M mask probability of a
 * proposal is zero-padded by `padding`, its box is expanded about the centre by (M+2p)/M (expand_boxes
 * :93-107), truncated to int32, the padded mask is resized to the box with bilinear interpolation
 * (torch F.interpolate, align_corners=False: src = fma(scale, dst+0.5, -0.5) clamped at 0, scale = in/out;
 * value = fma(ly0, fma(lx0, v00, lx1*v01), ly1 * fma(lx0, v10, lx1*v11)) -- the contraction pattern of torch's
 * vectorised CPU kernel; its scalar tail path differs in the last ulp, so goldens are matched to 2.4e-7) and
 * pasted into an im_h x im_w plane; the tight box of (plane > thresh) is returned as
 * [xmin, ymin, xmax, ymax] (inclusive), or [0, 0, im_h, im_w] if nothing passes (the reference's order).
 * Pinned by tests/golden/g9_paste.npz (the same steps executed with torch in the build container).
 * ---------------------------------------------------------------------------------- */
DMMO_API void dmmo_paste_mask(const float *prob, int M, const float *box, int im_h, int im_w, float thresh,
                              int padding, float *plane, float *new_box) {
    const int Mp = M + 2 * padding;
    float *pad = (float *)calloc((size_t)Mp * Mp, sizeof(float));
    for (int y = 0; y < M; ++y)
        for (int x = 0; x < M; ++x) pad[(y + padding) * Mp + x + padding] = prob[y * M + x];
    const float scale = (float)((double)Mp / (double)M);      /* python float, then fp32 tensor multiply */
    float w_half = (box[2] - box[0]) * 0.5f, h_half = (box[3] - box[1]) * 0.5f;
    const float x_c = (box[2] + box[0]) * 0.5f, y_c = (box[3] + box[1]) * 0.5f;
    w_half = w_half * scale;
    h_half = h_half * scale;
    const int bx0 = (int)(x_c - w_half), by0 = (int)(y_c - h_half);      /* .to(int32): truncation */
    const int bx1 = (int)(x_c + w_half), by1 = (int)(y_c + h_half);
    int w = bx1 - bx0 + 1, h = by1 - by0 + 1;
    if (w < 1) w = 1;
    if (h < 1) h = 1;
    memset(plane, 0, sizeof(float) * (size_t)im_h * im_w);
    const int x_0 = bx0 > 0 ? bx0 : 0, y_0 = by0 > 0 ? by0 : 0;
    const int x_1 = bx1 + 1 < im_w ? bx1 + 1 : im_w, y_1 = by1 + 1 < im_h ? by1 + 1 : im_h;
    const float sh = (float)Mp / (float)h, sw = (float)Mp / (float)w;
    int xmin = im_w, ymin = im_h, xmax = -1, ymax = -1;
    for (int y = y_0; y < y_1; ++y) {
        float ry = fmaf(sh, (float)(y - by0) + 0.5f, -0.5f);
        if (ry < 0.0f) ry = 0.0f;
        const int iy0 = (int)ry, iy1 = iy0 + (iy0 < Mp - 1 ? 1 : 0);
        const float ly1 = ry - (float)iy0, ly0 = 1.0f - ly1;
        for (int x = x_0; x < x_1; ++x) {
            float rx = fmaf(sw, (float)(x - bx0) + 0.5f, -0.5f);
            if (rx < 0.0f) rx = 0.0f;
            const int ix0 = (int)rx, ix1 = ix0 + (ix0 < Mp - 1 ? 1 : 0);
            const float lx1 = rx - (float)ix0, lx0 = 1.0f - lx1;
            const float t1 = lx1 * pad[iy0 * Mp + ix1], b1 = lx1 * pad[iy1 * Mp + ix1];
            const float top = fmaf(lx0, pad[iy0 * Mp + ix0], t1);
            const float bot = fmaf(lx0, pad[iy1 * Mp + ix0], b1);
            const float lb = ly1 * bot;
            const float v = fmaf(ly0, top, lb);
            plane[(size_t)y * im_w + x] = v;
            if (v > thresh) {
                if (x < xmin) xmin = x;
                if (x > xmax) xmax = x;
                if (y < ymin) ymin = y;
                if (y > ymax) ymax = y;
            }
        }
    }
    if (xmax < 0) { new_box[0] = 0; new_box[1] = 0; new_box[2] = (float)im_h; new_box[3] = (float)im_w; }
    else { new_box[0] = (float)xmin; new_box[1] = (float)ymin; new_box[2] = (float)xmax; new_box[3] = (float)ymax; }
    free(pad);
}

/* ------------------------------------------------------------------------------------
 * NMS as used by filter_results (dmm/utils/boxlist_ops.py:15-29): maskrcnn_benchmark.layers.nms
 * (third-party, un-pinned: greedy suppression in descending score order with the legacy "+1" box area,
 * IoU > thresh suppresses; restated from its published nms.cu).  Ties in score keep the lower index first.
 * keep_out receives the kept indices in score order; returns their count (capped at max_keep if > 0).
 * ---------------------------------------------------------------------------------- */
DMMO_API int dmmo_nms(const float *boxes, const float *scores, int n, float thresh, int max_keep, int32_t *keep_out) {
    int32_t *order = (int32_t *)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
    uint8_t *dead = (uint8_t *)calloc(n > 0 ? n : 1, 1);
    for (int i = 0; i < n; ++i) order[i] = i;
    for (int i = 1; i < n; ++i) {                                   /* stable insertion sort, descending */
        const int32_t k = order[i];
        int j = i - 1;
        while (j >= 0 && scores[order[j]] < scores[k]) { order[j + 1] = order[j]; --j; }
        order[j + 1] = k;
    }
    int cnt = 0;
    for (int a = 0; a < n; ++a) {
        const int i = order[a];
        if (dead[i]) continue;
        keep_out[cnt++] = i;
        if (max_keep > 0 && cnt >= max_keep) break;
        const float *bi = boxes + 4 * i;
        const float ai = (bi[2] - bi[0] + 1.0f) * (bi[3] - bi[1] + 1.0f);
        for (int c = a + 1; c < n; ++c) {
            const int j = order[c];
            if (dead[j]) continue;
            const float *bj = boxes + 4 * j;
            const float l = fmaxf(bi[0], bj[0]), r = fminf(bi[2], bj[2]);
            const float t = fmaxf(bi[1], bj[1]), bb = fminf(bi[3], bj[3]);
            const float iw = fmaxf(r - l + 1.0f, 0.0f), ih = fmaxf(bb - t + 1.0f, 0.0f);
            const float inter = iw * ih;
            const float aj = (bj[2] - bj[0] + 1.0f) * (bj[3] - bj[1] + 1.0f);
            if (inter / (ai + aj - inter) > thresh) dead[j] = 1;
        }
    }
    free(order);
    free(dead);
    return cnt;
}

/* ----------------------------------------------------------------------------------
 * Frame-loop reductions.
 * dmmo_mask_box: ohw_mask2boxlist for ONE plane (reference dmm/utils/utils.py:179-210 +
 * binmask_to_bbox_xyxy_pt :114-143): box = [xmin, ymin, xmax, ymax] of (mask > thresh) (the reference uses
 * thresh = 0: `mask > 0`, :120), the whole frame [0,0,W-1,H-1] when nothing passes (:195-196);
 * returns template_valid = (sum of the plane > 0) (:191), summed in double here (masks are non-negative).
 * ---------------------------------------------------------------------------------- */
DMMO_API int dmmo_mask_box(const float *mask, int H, int W, float thresh, float *box) {
    int xmin = W, ymin = H, xmax = -1, ymax = -1;
    double s = 0.0;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const float v = mask[(size_t)y * W + x];
            s += (double)v;
            if (v > thresh) {
                if (x < xmin) xmin = x;
                if (x > xmax) xmax = x;
                if (y < ymin) ymin = y;
                if (y > ymax) ymax = y;
            }
        }
    if (xmax < 0) { box[0] = 0.0f; box[1] = 0.0f; box[2] = (float)(W - 1); box[3] = (float)(H - 1); }
    else { box[0] = (float)xmin; box[1] = (float)ymin; box[2] = (float)xmax; box[3] = (float)ymax; }
    return s > 0.0;
}

/* dmmo_merge_labels: label map of one video frame (reference dmm/modules/evaluator.py:134-139):
 * refine_bg = 1 - max_o mask[o]; labels = argmax over cat([bg, mask[0..O)]) along dim 0; torch CPU max(dim)
 * returns the first maximal index.  O == 0 -> all background (the reference never reaches this case). */
DMMO_API void dmmo_merge_labels(const float *masks, int O, int HW, uint8_t *labels) {
    for (int x = 0; x < HW; ++x) {
        if (O <= 0) { labels[x] = 0; continue; }
        float mx = masks[x];
        for (int o = 1; o < O; ++o) {
            const float v = masks[(size_t)o * HW + x];
            if (v > mx) mx = v;
        }
        float best = 1.0f - mx;
        int arg = 0;
        for (int o = 0; o < O; ++o) {
            const float v = masks[(size_t)o * HW + x];
            if (v > best) { best = v; arg = o + 1; }
        }
        labels[x] = (uint8_t)arg;
    }
}
