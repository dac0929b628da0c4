// dmm_solve_rs.hip -- ROW-SPLIT form of the relaxed-assignment solver (same arithmetic as dmm_solve.hip, other mapping).
//
// Reference: relax_matching / project_row / project_col  dmm/modules/submodules/relax_match.py:9-105,
//            sim mix + pad + mean(X_list) + logic + scores  dmm/modules/match_model.py:89-130, :146-147.
//
// dmm_solve.hip gives one thread a whole COLUMN (all M rows in registers): ideal when thousands of frames are in
// flight (one wave per frame on every SIMD), but a single frame is then one wave issuing ~370 dependent-ish VALU
// instructions per sweep (1.0 us / sweep at 10 x 50).  Here a frame is solved by a WORKGROUP OF RG x CG WAVES
// ("rows across waves, column sums via LDS"):
//
//     wave (rg, cg): rows [rg*RW, rg*RW + RW) x columns [cg*64, cg*64 + 64);  lane = column, RW <= 8 rows in registers.
//
// State per thread: RW x (C, X, P0, P1, P2, acc).  The reductions go through LDS:
//   * column sums: every wave reads its lanes' columns (all n rows) from the ping-pong buffer the relu step wrote;
//   * row sums:    the waves of a row group read their own rows (aligned 8-lane groups = ATen's AVX2 lanes).
// One barrier per sweep (two when CG > 1).  The "did anything move" flags of sweep j are read at the barrier of sweep
// j+1 (the relu step done in between is undone when the reference would have left the loop), so the exit test costs
// no barrier of its own.
//
// RESULT (tools/solver_timing.py, MI355X, us per solve at 20 x 5 iterations, thread-per-column vs this): 5 x 50: 61 vs
// 82; 10 x 50: 102 vs 99; 16 x 64: 124 vs 112; 20 x 200 (CG = 4): 216 vs 343.  The sweep is bound by instruction issue,
// not by the row count: the column-sum and row-sum code is replicated in every wave and the per-row work is the small
// part (~300 instructions per sweep and wave at RW = 3 against ~370 for all 10 rows in one wave), and each sweep adds
// a barrier and three LDS round trips.  So this mapping is used only where it wins (>= 12 rows, <= 64 columns, few
// frames in flight; dmm_solve.hip use_row_split), and only the CG = 1 form is instantiated.
//
// BIT EXACTNESS: every fp32 operation and every summation order is the one of dmm_solve.hip / dmm_torch_order.h
// (ATen outer-sum order for the column sums, vectorized inner-sum order for the row sums, the 2-norm fast path for
// the cost); only the placement of the operands differs.  tests/test_gpu_parity.py runs all solver goldens through
// both kernels (option DMM_OPT_SOLVER_KERNEL = 0 | 1).
#include "dmm_solve.h"

namespace dmm {
namespace rs {

constexpr int kMaxWaves = 16;
constexpr int kFoldStride = DMM_MAX_TEMPLATES;

// LDS pointers carry their address space in the type: through generic pointers (a struct of float*, indexed
// dynamically) the compiler fell back to flat_load / flat_store + a scratch-resident pointer table, 3x slower per sweep.
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) int lds_i32;

// LDS layout.  Three [4 RW][LD] fp32 matrices with a COMPILE-TIME leading dimension LD = 64 CG (lane = column, so row
// i of a thread's column is an immediate offset i*LD*4 from one per-lane address: no address arithmetic in the sweep):
//   xb[0], xb[1]  ping-pong: X after the relu step (the column sums read all rows of a column)
//   yb            X after the column projection (the row sums read the wave's own rows)
// Rows >= n and columns >= m hold exact zeros at all times (their X never leaves 0), so the column sums can add every
// one of the 4 RW rows without guards.  The cost products need the reference's DENSE [n, m] order for the 2-norm: they
// are written with row stride m over the idle ping-pong buffer (it is rewritten in full before it is read again).
template <int RW, int CG>
struct Smem {
    static constexpr int LD = 64 * CG;
    static constexpr int ROWS = 4 * RW;
    static constexpr int SZ = ROWS * LD + 64;          // + 64: the batched tail reads of the last row may run past it
    lds_f32 *base;
    __device__ __forceinline__ lds_f32 *xb(int pp) const { return base + pp * SZ; }
    __device__ __forceinline__ lds_f32 *yb() const { return base + 2 * SZ; }
    __device__ __forceinline__ lds_f32 *fold() const { return base + 3 * SZ; }       // [kMaxWaves][kFoldStride]
    __device__ __forceinline__ lds_i32 *flags(int pp) const {                          // [2][kMaxWaves]
        return (lds_i32 *)(base + 3 * SZ + kMaxWaves * kFoldStride) + pp * kMaxWaves;
    }
    __device__ __forceinline__ lds_f32 *cost() const { return base + 3 * SZ + kMaxWaves * kFoldStride + 2 * kMaxWaves; }
    static constexpr size_t bytes() { return sizeof(float) * (3 * SZ + kMaxWaves * kFoldStride + 2 * kMaxWaves + 16); }
};

__device__ __forceinline__ void lds_wave_sync() {
    // LDS operations of one wave execute in order; this only stops the compiler from moving accesses across and
    // drains the queue so that values written by other LANES of this wave are visible to the following reads.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// fold RW wave-uniform values over the CG waves of a row group (prologue / epilogue: two barriers)
template <int RW, int CG, typename OP>
__device__ __forceinline__ void fold_cg(float (&v)[RW], lds_f32 *fold, int cg, int r0, OP op) {
    if (CG == 1) return;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < RW; ++k) fold[cg * kFoldStride + ((r0 + k) & (kFoldStride - 1))] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RW; ++k) {
        float t = fold[(r0 + k) & (kFoldStride - 1)];
#pragma unroll
        for (int c = 1; c < CG; ++c) t = op(t, fold[c * kFoldStride + ((r0 + k) & (kFoldStride - 1))]);
        v[k] = t;
    }
    __syncthreads();
}
template <int RW, int CG>
__device__ __forceinline__ void fold_cg_min_i32(int (&v)[RW], lds_f32 *fold_f, int cg, int r0) {
    if (CG == 1) return;
    lds_i32 *fold = (lds_i32 *)fold_f;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < RW; ++k) fold[cg * kFoldStride + ((r0 + k) & (kFoldStride - 1))] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RW; ++k) {
        int t = fold[(r0 + k) & (kFoldStride - 1)];
#pragma unroll
        for (int c = 1; c < CG; ++c) {
            const int o = fold[c * kFoldStride + ((r0 + k) & (kFoldStride - 1))];
            t = o < t ? o : t;
        }
        v[k] = t;
    }
    __syncthreads();
}

// Row sums of this wave's RW rows of buf ([.., LD]; m live columns) in ATen's vectorized inner-sum order: the aligned
// 8-lane group g of the wave sums row r0 + g; results come back wave-uniform in rs[].
template <int RW, int CG>
__device__ __forceinline__ void row_sums_own(const lds_f32 *buf, int m, int r0, float (&rs)[RW]) {
    constexpr int LD = 64 * CG;
    const int lane = threadIdx.x & 63, l = lane & 7, g = lane >> 3;
    const int rr = g < RW ? r0 + g : r0;                        // groups beyond the wave's rows recompute row r0
    const lds_f32 *x = buf + rr * LD;
    float s;
    if (CG == 1) {
        // m <= 64: every LDS word a group can need -- 8 vector slots x[8 i + l] and up to 7 tail scalars -- is fetched
        // in one batch, the adds then follow the ATen order under wave-uniform conditions on m (dmm_solve.hip,
        // row_sums_torch_order_wave).  Words past column m are zeros or the next row: never used.
        const int vs = m >> 3, gq = vs >> 2, tail0 = vs << 3, ntail = m - tail0;
        float v[8], t[7];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = x[8 * i + l];
#pragma unroll
        for (int k = 0; k < 7; ++k) t[k] = x[tail0 + k];
        if (m < torder::TV) {
            float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
            if (m >= 4) { p0 = p0 + t[0]; p1 = p1 + t[1]; p2 = p2 + t[2]; p3 = p3 + t[3]; }
            const int b4 = m & ~3;
#pragma unroll
            for (int k = 0; k < 7; ++k)
                if (k >= b4 && k < m) p0 = p0 + t[k];
            p0 = p0 + p1; p0 = p0 + p2; p0 = p0 + p3;
            s = p0;
        } else {
            float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
            if (gq >= 1) { p0 = p0 + v[0]; p1 = p1 + v[1]; p2 = p2 + v[2]; p3 = p3 + v[3]; }
            if (gq >= 2) { p0 = p0 + v[4]; p1 = p1 + v[5]; p2 = p2 + v[6]; p3 = p3 + v[7]; }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i >= 4 * gq && i < vs) p0 = p0 + v[i];
            p0 = p0 + p1; p0 = p0 + p2; p0 = p0 + p3;
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 7; ++k)
                if (k < ntail) acc = acc + t[k];
            s = torder::add_group8_seq(acc, p0);
        }
    } else {
        s = torder::inner_sum_group8_small(m, l, [&](int i) { return x[i]; });      // m <= 256
    }
#pragma unroll
    for (int k = 0; k < RW; ++k) rs[k] = readlane_f32(s, 8 * k);
}

// X.sum(dim=0) of one column (xcol = buffer + column; row i at xcol[i*LD]) in ATen's outer-sum order for the column's
// class; n <= MT live rows, rows >= n hold zeros (adding them is exact; only the ILP-4 class needs to know n: its
// remainder rows n4 .. n-1 join chain 0).  All MT words are fetched before the first add.
template <int MT, int LD>
__device__ __forceinline__ void col_fetch(const lds_f32 *xcol, float (&v)[MT]) {
#pragma unroll
    for (int i = 0; i < MT; ++i) v[i] = xcol[i * LD];
}
template <int MT>
__device__ __forceinline__ float col_sum(const float (&v)[MT], int n, bool class_a) {
    const int g = n >> 2;
    float a0 = 0.0f, a1 = 0.0f;
    float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        a0 = a0 + v[i];
        if (((i + 1) & 15) == 0) { a1 = a1 + a0; a0 = 0.0f; }
        const float xm = (i >> 2) < g ? v[i] : 0.0f;            // one wave-uniform condition per group of 4 rows
        if ((i & 3) == 0) p0 = p0 + xm;
        else if ((i & 3) == 1) p1 = p1 + xm;
        else if ((i & 3) == 2) p2 = p2 + xm;
        else p3 = p3 + xm;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) p0 = p0 + ((i >> 2) < g ? 0.0f : v[i]);
    p0 = p0 + p1;
    p0 = p0 + p2;
    p0 = p0 + p3;
    return class_a ? a0 + a1 : p0;
}

// ---------------------------------------------------------------------------------------------------------------
// relax_matching core, row-split.  C[k] = cost of (row r0 + k, this thread's column), zeros outside the live [n, m]
// block.  n <= 4 RW, m <= 64 CG.  The three LDS matrices must be ZERO on entry.  On return X[] is the final projected
// iterate of the thread's tile, acc[] = sum(X_list); returns len(X_list) - 1.  Called by every thread of the block.
// ---------------------------------------------------------------------------------------------------------------
template <int RW, int CG>
__device__ __forceinline__ int relax_core_rs(const float (&C)[RW], int n, int m, int col, int r0, int wave, int nwaves,
                                             int cg, const RelaxParams prm, const Smem<RW, CG> &sm, float (&X)[RW],
                                             float (&acc)[RW], float *cost_out) {
    constexpr int MT = 4 * RW, LD = 64 * CG;
    const int lane = threadIdx.x & 63;
    const bool live = col < m;
    const float fn = (float)n, fm = (float)m;
    const float rcp_n = 1.0f / fn, rcp_m = 1.0f / fm;
    const bool col_class_a = col < torder::outer_class_bound(m);
    const int tile = r0 * LD + col;                              // this thread's (row r0, column) word in every matrix
#define RS_ROW(k) ((r0 + (k)) < n)

    // ---- greedy row-min initialisation (relax_match.py:45-55); max / first-argmin are order free ----
    float cm = -__builtin_inff();
#pragma unroll
    for (int k = 0; k < RW; ++k)
        if (RS_ROW(k) && live) cm = C[k] > cm ? C[k] : cm;
    cm = wave_max(cm);
    if (lane == 0) sm.fold()[wave] = cm;
#pragma unroll
    for (int k = 0; k < RW; ++k) sm.xb(0)[tile + k * LD] = C[k];
    __syncthreads();
    // every lane reads one wave's partial (lanes >= nwaves re-read the last one): one LDS round trip, then a wave max
    const float cmax = wave_max(sm.fold()[lane < nwaves ? lane : nwaves - 1]);
    int best_row = 0;
    {
        float cv[MT];
        col_fetch<MT, LD>(sm.xb(0) + col, cv);
        float bv = cv[0];
#pragma unroll
        for (int i = 1; i < MT; ++i)
            if (i < n && cv[i] < bv) { bv = cv[i]; best_row = i; }      // first argmin over the rows of this column
    }
    __syncthreads();                                            // fold[] and xb[0] are reused below
    {
        float crm[RW], vmin[RW];
        int cand[RW];
#pragma unroll
        for (int k = 0; k < RW; ++k) {
            crm[k] = (live && RS_ROW(k)) ? ((r0 + k) == best_row ? C[k] : cmax) : __builtin_inff();
            vmin[k] = crm[k];
        }
        wave_min_rows<RW>(vmin);
        fold_cg<RW, CG>(vmin, sm.fold(), cg, r0, op_fmin());
#pragma unroll
        for (int k = 0; k < RW; ++k) cand[k] = (live && crm[k] == vmin[k]) ? col : 0x7fffffff;
        wave_min_rows_i32<RW>(cand);
        fold_cg_min_i32<RW, CG>(cand, sm.fold(), cg, r0);       // first argmin over the columns
#pragma unroll
        for (int k = 0; k < RW; ++k) X[k] = (RS_ROW(k) && col == cand[k]) ? 1.0f : 0.0f;
    }
    float P0[RW], P1[RW], P2[RW];
#pragma unroll
    for (int k = 0; k < RW; ++k) {
        P0[k] = 0.0f; P1[k] = 0.0f; P2[k] = 0.0f;
        acc[k] = 0.0f + X[k];
    }
    if (cost_out && threadIdx.x == 0) cost_out[0] = 0.0f;

    int len = 1, pp = 0;
    float cost_prev = 0.0f;
    for (int it = 0; it < prm.max_iter; ++it) {
        // gradient step X = X - lr*C (:69); cost = ||X*C||_F (:70); X_list.append(X) (:71)
#pragma unroll
        for (int k = 0; k < RW; ++k) {
            const float g = prm.lr * C[k];
            X[k] = X[k] - g;
            acc[k] = acc[k] + X[k];
        }
        lds_f32 *cb = sm.xb(pp ^ 1);                            // dense [n, m] products over the idle ping-pong buffer
        if (live) {
#pragma unroll
            for (int k = 0; k < RW; ++k)
                if (RS_ROW(k)) cb[(r0 + k) * m + col] = X[k] * C[k];
        }
        __syncthreads();                                        // S0: the products are complete
        if (threadIdx.x < 8) {                                  // wave 0, one aligned 8-lane group: ATen's 2-norm order
            const float c = torder::norm2_group8((long)n * m, (int)threadIdx.x, [&](long i) { return cb[(int)i]; });
            if (threadIdx.x == 0) {
                *sm.cost() = c;
                if (cost_out) cost_out[it + 1] = c;
            }
        }
        ++len;

        for (int j = 0; j < prm.proj_iter; ++j) {
            float Xs[RW], P0s[RW];
            // {X >= 0} (:74-76) then X = Y + P1 (:78)
#pragma unroll
            for (int k = 0; k < RW; ++k) {
                Xs[k] = X[k];
                P0s[k] = P0[k];
                const float x = X[k] + P0[k];
                const float y = x > 0.0f ? x : 0.0f;
                P0[k] = x - y;
                X[k] = y + P1[k];
            }
            // (unconditional stores: the dead columns' zeros also wipe what this iteration's dense products left)
#pragma unroll
            for (int k = 0; k < RW; ++k) sm.xb(pp)[tile + k * LD] = X[k];
            __syncthreads();                                    // S_A: xb[pp] complete; flags of sweep j-1 visible
            float cv[MT];
            col_fetch<MT, LD>(sm.xb(pp) + col, cv);
            if (j > 0) {
                // if ||X - X_start|| == 0: break (:88-89), decided one barrier late: sweep j-1 moved nothing anywhere
                // -> it was the last one; take back this sweep's relu step
                const int fl = sm.flags(pp ^ 1)[lane < nwaves ? lane : nwaves - 1];   // one flag per lane, one round trip
                if (__ballot(fl != 0) == 0ull) {
#pragma unroll
                    for (int k = 0; k < RW; ++k) { X[k] = Xs[k]; P0[k] = P0s[k]; }
                    break;
                }
            }
            // {column sums <= 1}: project_col (:21-34, :79-80); then X = Y + P2 (:82)
            const float cs = col_sum<MT>(cv, n, col_class_a);
            const bool over = cs > 1.0f;                        // dead columns hold zeros: never over
            const float tc = div_by_const(cs - 1.0f, fn, rcp_n);
#pragma unroll
            for (int k = 0; k < RW; ++k) {
                float x = X[k];
                const float tci = RS_ROW(k) ? tc : 0.0f;
                const float y = over ? x - tci : x;
                P1[k] = x - y;
                x = y + P2[k];
                X[k] = x;
            }
#pragma unroll
            for (int k = 0; k < RW; ++k) sm.yb()[tile + k * LD] = X[k];
            if (CG > 1) __syncthreads();                        // S_B: a row spans CG waves
            else lds_wave_sync();
            // {row sums = 1}: project_row (:9-19, :83-84)
            float rsv[RW];
            row_sums_own<RW, CG>(sm.yb(), m, r0, rsv);
            unsigned moved_bits = 0;
#pragma unroll
            for (int k = 0; k < RW; ++k) {
                float tr = div_by_const(rsv[k] - 1.0f, fm, rcp_m);
                tr = (live && RS_ROW(k)) ? tr : 0.0f;
                const float x = X[k];
                const float y = x - tr;
                P2[k] = x - y;
                X[k] = y;                                       // :86
                const float d = y - Xs[k];
                const float sq = d * d;
                moved_bits |= __float_as_uint(sq);
            }
            const bool wave_moved = __ballot(moved_bits != 0u) != 0ull;
            if (lane == 0) sm.flags(pp)[wave] = wave_moved ? 1 : 0;
            pp ^= 1;
        }
        __syncthreads();                                        // S_C: the cost of this iteration is visible
        const float cost = *sm.cost();
        if (cost_prev == cost) break;                           // :96-98
        cost_prev = cost;
    }
#undef RS_ROW
    return len - 1;
}

template <int RW, int CG>
__device__ __forceinline__ Smem<RW, CG> carve(float *smem) {
    Smem<RW, CG> s;
    s.base = (lds_f32 *)smem;
    // zero the three matrices: rows >= n / columns >= m are read (never written with anything but zeros)
    for (int i = threadIdx.x; i < 3 * Smem<RW, CG>::SZ; i += blockDim.x) s.base[i] = 0.0f;
    __syncthreads();
    return s;
}

// ---------------------------------------------------------------------------------------------------------------
// Layer kernel (same contract as relax_match_kernel in dmm_solve.hip).  grid = B, block = 64 * RG * CG.
// ---------------------------------------------------------------------------------------------------------------
template <int RW, int CG>
__global__ __launch_bounds__(256 * CG) void relax_match_rs_kernel(
    const float *__restrict__ cos_in, const int32_t *__restrict__ inter, const int32_t *__restrict__ area_p,
    const int32_t *__restrict__ area_t, const float *__restrict__ score_p, int N, int M,
    const int32_t *__restrict__ n_valid, const int32_t *__restrict__ m_valid, float w_feat, float w_iou,
    RelaxParams prm, int is_test, float *__restrict__ sim_out, float *__restrict__ R_out, float *__restrict__ Rb_out,
    float *__restrict__ match_score, float *__restrict__ det_score, int32_t *__restrict__ iters_out,
    float *__restrict__ X_final) {
    extern __shared__ float smem_rs[];
    const int b = blockIdx.x;
    const int nthreads = blockDim.x, nwaves = nthreads >> 6;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cg = wave % CG, rg = wave / CG;
    const int col = cg * 64 + lane, r0 = rg * RW;
    const int Nb = n_valid ? n_valid[b] : N;
    const int Mb = m_valid ? m_valid[b] : M;
    const int PpS = N > M ? N : M + 1;
    float *Rb_b = Rb_out + (int64_t)b * M * PpS;
    float *R_b = R_out ? R_out + (int64_t)b * M * PpS : nullptr;
    float *X_b = X_final ? X_final + (int64_t)b * M * PpS : nullptr;
    float *sim_b = sim_out + (int64_t)b * M * N;
    if (Mb <= 0 || Nb <= 0) {                                   // dead frame: zeros (dmm_model.py:118-122)
        for (int i = threadIdx.x; i < M * PpS; i += nthreads) {
            Rb_b[i] = 0.0f;
            if (R_b) R_b[i] = 0.0f;
            if (X_b) X_b[i] = 0.0f;
        }
        for (int i = threadIdx.x; i < M * N; i += nthreads) sim_b[i] = 0.0f;
        for (int i = threadIdx.x; i < M; i += nthreads) {
            match_score[(int64_t)b * M + i] = 0.0f;
            det_score[(int64_t)b * M + i] = 0.0f;
        }
        if (iters_out && threadIdx.x == 0) iters_out[b] = 0;
        return;
    }
    const int Pp = Nb > Mb ? Nb : Mb + 1;                       // live solver width (match_model.py:109-113)
    const bool has_prop = col < Nb;
    const Smem<RW, CG> sm = carve<RW, CG>(smem_rs);
#define RS_ROW(k) ((r0 + (k)) < Mb)

    // ---- sim = (1-w)*cos + w*iou; pad; C = -sim ----
    float C[RW];
    {
        const float *cos_b = cos_in + (int64_t)b * M * N;
        const int32_t *inter_b = inter + (int64_t)b * M * N;
        const int ap = has_prop ? area_p[(int64_t)b * N + col] : 0;
#pragma unroll
        for (int k = 0; k < RW; ++k) {
            const int i = r0 + k;
            float simv = 0.0f;
            C[k] = 0.0f;
            if (RS_ROW(k) && has_prop) {
                const int in = inter_b[(int64_t)i * N + col];
                const int un = ap + area_t[(int64_t)b * M + i] - in;
                const float iou = (float)in / ((float)un + 1e-6f);     // match_helper.py:24-27
                const float a = cos_b[(int64_t)i * N + col] * w_feat, c = iou * w_iou;
                simv = a + c;                                          // match_model.py:90
                sim_b[(int64_t)i * N + col] = simv;
            }
            if (RS_ROW(k) && col < Pp) C[k] = -simv;                   // padded columns: -0.0
        }
    }
    float X[RW], acc[RW];
    const int iters = relax_core_rs<RW, CG>(C, Mb, Pp, col, r0, wave, nwaves, cg, prm, sm, X, acc, nullptr);
    if (iters_out && threadIdx.x == 0) iters_out[b] = iters;

    // ---- R = sum(X_list)/len; logic; Rb; scores ----
    const float flen = (float)(iters + 1);
    const float sc = has_prop ? score_p[(int64_t)b * N + col] : 0.0f;
    const bool livec = col < Pp;
    float r[RW], rmax[RW], ms[RW];
#pragma unroll
    for (int k = 0; k < RW; ++k) {
        r[k] = acc[k] / flen;                                          // match_model.py:121
        rmax[k] = (livec && RS_ROW(k)) ? r[k] : -__builtin_inff();
    }
    wave_max_rows<RW>(rmax);
    fold_cg<RW, CG>(rmax, sm.fold(), cg, r0, op_fmax());
#pragma unroll
    for (int k = 0; k < RW; ++k) {
        const int i = r0 + k;
        const float lg = is_test ? (r[k] == rmax[k] ? 1.0f : 0.0f) : (r[k] > 0.01f ? 1.0f : 0.0f);
        const float rb = (livec && RS_ROW(k)) ? r[k] * lg : 0.0f;      // :130
        const float rc = r[k] < 0.0f ? 0.0f : (r[k] > 1.0f ? 1.0f : r[k]);
        ms[k] = (livec && RS_ROW(k)) ? rc * (-C[k]) : -__builtin_inff();   // :146
        const float ds = sc * rb;                                      // :147
        sm.yb()[i * (64 * CG) + col] = (RS_ROW(k) && livec) ? ds : 0.0f;
        if (RS_ROW(k) && col < PpS) {
            Rb_b[(int64_t)i * PpS + col] = rb;
            if (R_b) R_b[(int64_t)i * PpS + col] = livec ? r[k] : 0.0f;
            if (X_b) X_b[(int64_t)i * PpS + col] = livec ? X[k] : 0.0f;
        }
    }
    wave_max_rows<RW>(ms);
    fold_cg<RW, CG>(ms, sm.fold(), cg, r0, op_fmax());
    __syncthreads();                                                   // yb (score * Rb) complete
    float dsum[RW];
    row_sums_own<RW, CG>(sm.yb(), Pp, r0, dsum);                     // (score * Rb).sum(1), ATen inner-sum order
    if (lane == 0 && cg == 0) {
#pragma unroll
        for (int k = 0; k < RW; ++k)
            if (RS_ROW(k)) {
                match_score[(int64_t)b * M + r0 + k] = ms[k];
                det_score[(int64_t)b * M + r0 + k] = dsum[k];
            }
    }
#undef RS_ROW
    // rows of dead templates, dead proposal columns of sim: zeros (cooperatively, any thread)
    for (int i = Mb * PpS + threadIdx.x; i < M * PpS; i += nthreads) {
        Rb_b[i] = 0.0f;
        if (R_b) R_b[i] = 0.0f;
        if (X_b) X_b[i] = 0.0f;
    }
    for (int i = Mb * N + threadIdx.x; i < M * N; i += nthreads) sim_b[i] = 0.0f;
    for (int i = Mb + threadIdx.x; i < M; i += nthreads) {
        match_score[(int64_t)b * M + i] = 0.0f;
        det_score[(int64_t)b * M + i] = 0.0f;
    }
    if (Nb < N)
        for (int i = threadIdx.x; i < Mb * (N - Nb); i += nthreads) sim_b[(i / (N - Nb)) * N + Nb + i % (N - Nb)] = 0.0f;
    // live rows, columns in [Pp, PpS): zeros (the thread tiles cover col < 64*CG >= PpS, handled above via col < PpS)
}

// Solver-only kernel on a caller-provided C [B, n, m] (same contract as relax_solve_kernel).
template <int RW, int CG>
__global__ __launch_bounds__(256 * CG) void relax_solve_rs_kernel(const float *__restrict__ Cin, int n_max,
                                                                         int m_max,
                                                                         const int32_t *__restrict__ rows_valid,
                                                                         const int32_t *__restrict__ cols_valid,
                                                                         RelaxParams prm, float *__restrict__ X_final,
                                                                         float *__restrict__ R_out,
                                                                         float *__restrict__ cost_out,
                                                                         int32_t *__restrict__ iters_out) {
    extern __shared__ float smem_rs[];
    const int b = blockIdx.x;
    const int nthreads = blockDim.x, nwaves = nthreads >> 6;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cg = wave % CG, rg = wave / CG;
    const int col = cg * 64 + lane, r0 = rg * RW;
    const int n = rows_valid ? rows_valid[b] : n_max;
    const int m = cols_valid ? cols_valid[b] : m_max;
    if (n <= 0 || m <= 0) {                                     // dead frame
        for (int i = threadIdx.x; i < n_max * m_max; i += nthreads) {
            if (X_final) X_final[(int64_t)b * n_max * m_max + i] = 0.0f;
            if (R_out) R_out[(int64_t)b * n_max * m_max + i] = 0.0f;
        }
        if (iters_out && threadIdx.x == 0) iters_out[b] = 0;
        return;
    }
    const Smem<RW, CG> sm = carve<RW, CG>(smem_rs);
    float C[RW], X[RW], acc[RW];
#pragma unroll
    for (int k = 0; k < RW; ++k)
        C[k] = (r0 + k < n && col < m) ? Cin[((int64_t)b * n_max + r0 + k) * m_max + col] : 0.0f;
    const int iters = relax_core_rs<RW, CG>(C, n, m, col, r0, wave, nwaves, cg, prm, sm, X, acc,
                                            cost_out ? cost_out + (int64_t)b * (prm.max_iter + 1) : nullptr);
    const float flen = (float)(iters + 1);
#pragma unroll
    for (int k = 0; k < RW; ++k) {
        const int i = r0 + k;
        if (i < n_max && col < m_max) {
            const bool lv = i < n && col < m;
            if (X_final) X_final[((int64_t)b * n_max + i) * m_max + col] = lv ? X[k] : 0.0f;
            if (R_out) R_out[((int64_t)b * n_max + i) * m_max + col] = lv ? acc[k] / flen : 0.0f;
        }
    }
    if (iters_out && threadIdx.x == 0) iters_out[b] = iters;
}

}  // namespace rs

// ---- launchers --------------------------------------------------------------------------------------------------
// Only the one-column-group form (Pp <= 64) is instantiated: with CG = 2 / 4 (two barriers per sweep, 8-16 waves) the
// row-split mapping measured 1.5x SLOWER than thread-per-column (20 x 200: 343 vs 216 us, 32 x 256: 662 vs 448 us).
#define DMM_RS_DISPATCH(RW_, CG_, CALL)                                                                            \
    do {                                                                                                           \
        switch ((RW_)) {                                                                                           \
            case 1: CALL(1, 1); break; case 2: CALL(2, 1); break; case 3: CALL(3, 1); break;                       \
            case 4: CALL(4, 1); break; case 5: CALL(5, 1); break; case 6: CALL(6, 1); break;                       \
            case 7: CALL(7, 1); break; default: CALL(8, 1); break;                                                 \
        }                                                                                                          \
    } while (0)

struct RsShape {
    int RW, CG, RG;
};
static RsShape rs_shape(int rows, int width) {
    RsShape s;
    s.CG = 1;                                                   // width <= 64 (use_row_split)
    s.RW = (rows + 3) / 4;                                      // at most 4 row groups: n <= 4 RW inside the core
    s.RG = (rows + s.RW - 1) / s.RW;
    return s;
}

template <typename K>
static int rs_prepare(K kernel, size_t smem) {
    if (smem > 48 * 1024) DMM_HIP_TRY(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    return DMM_OK;
}

int launch_relax_match_rs(const float *cos_in, const int32_t *inter, const int32_t *area_p, const int32_t *area_t,
                          const float *score_p, int B, int N, int M, const int32_t *n_valid, const int32_t *m_valid,
                          float w_feat, float w_iou, RelaxParams prm, int is_test, float *sim_out, float *R_out,
                          float *Rb_out, float *match_score, float *det_score, int32_t *iters_out, float *X_final,
                          hipStream_t stream) {
    const int PpS = N > M ? N : M + 1;
    const RsShape s = rs_shape(M, PpS);
#define DMM_CALL(RW_, CG_)                                                                                           \
    do {                                                                                                             \
        const size_t smem_ = rs::Smem<RW_, CG_>::bytes();                                                            \
        int rc_ = rs_prepare(rs::relax_match_rs_kernel<RW_, CG_>, smem_);                                            \
        if (rc_ != DMM_OK) return rc_;                                                                               \
        hipLaunchKernelGGL((rs::relax_match_rs_kernel<RW_, CG_>), dim3(B), dim3(64 * s.RG * s.CG), smem_, stream,     \
                           cos_in, inter, area_p, area_t, score_p, N, M, n_valid, m_valid, w_feat, w_iou, prm,        \
                           is_test, sim_out, R_out, Rb_out, match_score, det_score, iters_out, X_final);              \
    } while (0)
    DMM_RS_DISPATCH(s.RW, s.CG, DMM_CALL);
#undef DMM_CALL
    return check_launch();
}

int launch_relax_solve_rs(const float *C, int B, int n, int m, const int32_t *rows_valid, const int32_t *cols_valid,
                          RelaxParams prm, float *X_final, float *R_out, float *cost_out, int32_t *iters_out,
                          hipStream_t stream) {
    const RsShape s = rs_shape(n, m);
#define DMM_CALL(RW_, CG_)                                                                                           \
    do {                                                                                                             \
        const size_t smem_ = rs::Smem<RW_, CG_>::bytes();                                                            \
        int rc_ = rs_prepare(rs::relax_solve_rs_kernel<RW_, CG_>, smem_);                                            \
        if (rc_ != DMM_OK) return rc_;                                                                               \
        hipLaunchKernelGGL((rs::relax_solve_rs_kernel<RW_, CG_>), dim3(B), dim3(64 * s.RG * s.CG), smem_, stream, C,  \
                           n, m, rows_valid, cols_valid, prm, X_final, R_out, cost_out, iters_out);                   \
    } while (0)
    DMM_RS_DISPATCH(s.RW, s.CG, DMM_CALL);
#undef DMM_CALL
    return check_launch();
}

}  // namespace dmm
