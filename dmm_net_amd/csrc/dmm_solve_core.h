// dmm_solve_core.h -- device code shared by the solver translation units (dmm_solve.hip: fp32 forms, bit exact;
// dmm_solve_h.hip: the fp16-state tolerance mode): reductions in ATen's order, the relax_matching cores and the layer
// body (prologue: sim mix + pad, epilogue: mean of iterates, logic, scores).  See dmm_solve.hip for the overview.
#pragma once
#include <stdlib.h>

#include "dmm_solve.h"

namespace dmm {


// ---------------------------------------------------------------------------------------------
// Cross-wave plumbing for NG > 1 (Pp > 64): per-wave partials go through LDS.
// ---------------------------------------------------------------------------------------------
template <int MT, int NG>
struct BlockRed {
    float *buf;  // [2][NG][MT + 1] floats (double buffered: one barrier per reduction)
    int phase;
    int wave;
    __device__ __forceinline__ BlockRed(float *b, int w) : buf(b), phase(0), wave(w) {}

    // vals[i] are wave-uniform partials; on return they hold the block totals (fixed wave order).
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
    // vals[r] = wave-uniform partial row sums; on return vals[r] = (block total of row r - 1) / fm, wave-uniform.  Lane r
    // of every wave adds the NG partials of row r and divides ONCE; the results come back through readlane.
    template <int CNT>
    __device__ __forceinline__ void row_steps(float (&vals)[CNT], float fm) {
        const int lane = threadIdx.x & 63;
        float t;
        if (NG == 1) {
            t = 0.0f;
#pragma unroll
            for (int i = 0; i < CNT; ++i) t = lane == i ? vals[i] : t;
        } else {
            float *p = buf + phase * NG * (MT + 1);
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < CNT; ++i) p[wave * (MT + 1) + i] = vals[i];
            }
            __syncthreads();
            const int r = lane < CNT ? lane : 0;
            t = p[r];
#pragma unroll
            for (int w = 1; w < NG; ++w) t = t + p[w * (MT + 1) + r];
            phase ^= 1;
        }
        t = (t - 1.0f) / fm;
#pragma unroll
        for (int i = 0; i < CNT; ++i) vals[i] = readlane_f32(t, i);
    }
    template <int CNT, typename OP>
    __device__ __forceinline__ void fold(float (&vals)[CNT], OP op) {
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
            for (int w = 1; w < NG; ++w) t = op(t, p[w * (MT + 1) + i]);
            vals[i] = t;
        }
        phase ^= 1;
    }
    template <int CNT>
    __device__ __forceinline__ void min_i32(int (&vals)[CNT]) {
        if (NG == 1) return;
        int *p = reinterpret_cast<int *>(buf + phase * NG * (MT + 1));
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int i = 0; i < CNT; ++i) p[wave * (MT + 1) + i] = vals[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
            int t = p[i];
#pragma unroll
            for (int w = 1; w < NG; ++w) { int o = p[w * (MT + 1) + i]; t = o < t ? o : t; }
            vals[i] = t;
        }
        phase ^= 1;
    }
};

struct fmax_op { __device__ __forceinline__ float operator()(float a, float b) const { return b > a ? b : a; } };
struct fmin_op { __device__ __forceinline__ float operator()(float a, float b) const { return b < a ? b : a; } };

// ---------------------------------------------------------------------------------------------
// Exact-order block reductions through LDS (dmm_torch_order.h).  xbuf holds an [n, m] matrix row-major
// with row stride m (exactly the reference tensor's layout); aligned 8-lane groups play the AVX2 lanes.
// Both routines contain their own barriers and must be called by every thread of the block.
// ---------------------------------------------------------------------------------------------
template <int NG>
__device__ __forceinline__ void row_sums_torch_order(const float *xbuf, int n, int m, float *rsbuf) {
    __syncthreads();                                   // xbuf complete
    const int l = threadIdx.x & 7;
    for (int r = threadIdx.x >> 3; r < n; r += 8 * NG) {
        const float *x = xbuf + r * m;
        // (a batched-read variant, torder::inner_sum_group8_batched, measured 25 % SLOWER per sweep here: at m = 200
        // it fetches 42 words where 25 are needed and the clamping arithmetic outweighs the saved round trips)
        const float s = torder::inner_sum_group8_small(m, l, [&](int i) { return x[i]; });   // m <= 256
        if (l == 0) rsbuf[r] = s;
    }
    __syncthreads();                                   // rsbuf complete, xbuf free again
}
// The row projection's step (row sum - 1) / m instead of the sum: computed ONCE per row by the lane that holds the sum
// (every thread recomputed it for every row before: 7 instructions x rows per sweep and thread).
template <int NG>
__device__ __forceinline__ void row_steps_torch_order(const float *xbuf, int n, int m, float fm, float rcp_m, float *rsbuf) {
    __syncthreads();                                   // xbuf complete
    const int l = threadIdx.x & 7;
    for (int r = threadIdx.x >> 3; r < n; r += 8 * NG) {
        const float *x = xbuf + r * m;
        const float s = torder::inner_sum_group8_small(m, l, [&](int i) { return x[i]; });   // m <= 256
        if (l == 0) rsbuf[r] = div_by_const(s - 1.0f, fm, rcp_m);
    }
    __syncthreads();                                   // rsbuf complete, xbuf free again
}
// Single-wave variant (NG == 1).  The rows live in their OWN buffer with a fixed stride of kRowStride = 72 floats: every
// lane writes its column (dead columns carry exact zeros) and slots 64..71 are zeroed once, so a row is zero-padded up
// to its stride.  That makes every address a compile-time constant and lets the tail of ATen's inner sum (the m % 8
// scalars behind the vectorised part) be 7 unconditional adds -- the slots past column m add exact zeros.  The shape
// of the vectorised part (m / 8 vectors: which go to the four ILP accumulators, which are appended to the first)
// is a compile-time parameter VS picked by ONE wave-uniform switch; a version with m as a run-time value throughout
// spent 60 of a sweep's 370 instructions on selects and read 15 words per row where 8 are needed (with m = 50 known
// at compile time the sweep is 239 instructions).  The 8 group results of a pass are picked out of the wave with
// v_readlane: rs[] comes back wave-uniform in registers.  Same add order as torder::inner_sum_group8_small.
constexpr int kRowStride = 72;

template <int MT, int VS>
__device__ __forceinline__ void row_sums_wave_vs(const float *rowbuf, int n, int m, float (&rs)[MT]) {
    const int lane = threadIdx.x & 63, l = lane & 7, g = lane >> 3;
#pragma unroll
    for (int p = 0; p < (MT + 7) / 8; ++p) {
        const int r = p * 8 + g;
        const float *x = rowbuf + (r < n ? r : n - 1) * kRowStride;
        float v[VS > 0 ? VS : 1], t[7];
#pragma unroll
        for (int i = 0; i < VS; ++i) v[i] = x[8 * i + l];
#pragma unroll
        for (int k = 0; k < 7; ++k) t[k] = x[8 * VS + k];        // zero from column m on
        float s;
        if (VS == 0) {                                 // scalar_inner_sum (m < 8): ILP-4 over single elements
            float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
            if (m >= 4) {
                p0 = p0 + t[0]; p1 = p1 + t[1]; p2 = p2 + t[2]; p3 = p3 + t[3];
                p0 = p0 + t[4]; p0 = p0 + t[5]; p0 = p0 + t[6];
            } else {
                p0 = p0 + t[0]; p0 = p0 + t[1]; p0 = p0 + t[2];
            }
            p0 = p0 + p1; p0 = p0 + p2; p0 = p0 + p3;
            s = p0;
        } else {
            constexpr int GQ = VS / 4;
            float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
            if (GQ >= 1) { p0 = p0 + v[0]; p1 = p1 + v[1]; p2 = p2 + v[2]; p3 = p3 + v[3]; }
            if (GQ >= 2) { p0 = p0 + v[4]; p1 = p1 + v[5]; p2 = p2 + v[6]; p3 = p3 + v[7]; }
#pragma unroll
            for (int i = 4 * GQ; i < VS; ++i) p0 = p0 + v[i];
            p0 = p0 + p1; p0 = p0 + p2; p0 = p0 + p3;   // vec[l]
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 7; ++k) acc = acc + t[k];
            s = torder::add_group8_seq(acc, p0);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (p * 8 + k < MT) rs[p * 8 + k] = readlane_f32(s, 8 * k);
    }
}
template <int MT>
__device__ __forceinline__ void row_sums_torch_order_wave(const float *rowbuf, int n, int m, float (&rs)[MT]) {
    __syncthreads();                                   // rowbuf complete (one wave: just drains the LDS queue)
    switch (m >> 3) {                                  // m <= 64 here (one wave)
        case 0: row_sums_wave_vs<MT, 0>(rowbuf, n, m, rs); break;
        case 1: row_sums_wave_vs<MT, 1>(rowbuf, n, m, rs); break;
        case 2: row_sums_wave_vs<MT, 2>(rowbuf, n, m, rs); break;
        case 3: row_sums_wave_vs<MT, 3>(rowbuf, n, m, rs); break;
        case 4: row_sums_wave_vs<MT, 4>(rowbuf, n, m, rs); break;
        case 5: row_sums_wave_vs<MT, 5>(rowbuf, n, m, rs); break;
        case 6: row_sums_wave_vs<MT, 6>(rowbuf, n, m, rs); break;
        case 7: row_sums_wave_vs<MT, 7>(rowbuf, n, m, rs); break;
        default: row_sums_wave_vs<MT, 8>(rowbuf, n, m, rs); break;
    }
}
__device__ __forceinline__ float norm_torch_order(const float *xbuf, int cnt, float *slot) {
    __syncthreads();
    if (threadIdx.x < 8) {
        const float c = torder::norm2_group8(cnt, threadIdx.x, [&](long i) { return xbuf[i]; });
        if (threadIdx.x == 0) *slot = c;
    }
    __syncthreads();
    return *slot;
}


// ---------------------------------------------------------------------------------------------
// relax_matching core, ONE WAVE per frame (NG == 1, forward only): the latency form of the sweep.  Same operations in
// the same order as relax_core below -- bit identical -- with the work a single wave has to ISSUE cut down, because for
// one wave per SIMD a sweep costs (instructions x 4 cycles) + a few fixed latencies:
//   * the element-wise Dykstra steps run on PAIRS of rows as packed fp32 (v_pk_add_f32 / v_pk_mul_f32: two IEEE ops
//     per instruction, separate roundings); the relu is one v_max_f32 (max(-0, +0) = +0 and max(NaN, 0) = 0 are what
//     `x > 0 ? x : 0` gives);
//   * the column projection subtracts `over ? tc : 0` (x - 0 = x exactly) instead of selecting per row;
//   * the row projection's (sum - 1) / m is computed ONCE per row, on the lanes that hold the row sums, before the
//     readlane broadcast (it was recomputed by every lane for every row: 7 instructions x rows per sweep); dead columns
//     are skipped under one exec mask instead of one select per row;
//   * "did anything move" = some |y - X_start| > 2^-75, i.e. exactly "some square (y - X_start)^2 is non-zero in fp32"
//     (d^2 rounds to zero iff |d| <= 2^-75; a NaN counts as moved), as one compare per row with the lane masks OR-ed on
//     the scalar unit -- no multiplies, no integer ORs;
//   * the trailing scalars of ATen's inner sum come as two 16-byte LDS reads per row instead of seven 4-byte ones, and
//     the rows are read back without draining the LDS queue first (one wave's LDS operations execute in order).
// C[i] = cost of (row i, this thread's column); n rows, m <= 64 columns live; threads with col >= m carry zeros.
// On return X[] is the final projected iterate, acc[] = sum(X_list); returns len(X_list) - 1.
// ---------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// Helper wave of the one-wave solver (workgroups of 128 threads, used while few frames are in flight).  Per outer
// iteration the reference evaluates cost = ||X * C||_F (relax_match.py:70) -- in ATen's order a chain of n*m/8 dependent
// fmas on 8 lanes, 0.6-0.9 us -- but only LOOKS at it after the projection sweeps (:96-98).  So wave 0 posts the products
// in LDS and goes on with its sweeps; wave 1 (another SIMD of the CU) computes the norm meanwhile and posts it back.
// Handshake through LDS words (one wave's LDS operations execute in order; workgroup-scope acquire / release):
//   hs[0] request: 0 = none yet, k = products of outer iteration k-1 are in xbuf (hs[3] = element count), -1 = stop
//   hs[1] done:    k = the cost of request k is in hs[2]
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int hs_load(const int *p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void hs_store(int *p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void norm_helper_wave(const float *xbuf, int *hs) {
    const int lane = threadIdx.x & 63;
    for (int seq = 1;; ++seq) {
        int f;
        while ((f = hs_load(&hs[0])) == seq - 1) __builtin_amdgcn_s_sleep(1);
        if (f < 0) return;
        if (lane < 8) {
            const float c = torder::norm2_group8(hs[3], lane, [&](long i) { return xbuf[i]; });
            if (lane == 0) hs[2] = __float_as_int(c);
        }
        hs_store(&hs[1], seq);
    }
}
// Kernel prologue for the one-wave forms: clears the handshake words; in a 128-thread workgroup the second wave becomes
// the helper and never returns to the caller's code (returns true: the caller must `return`).
__device__ __forceinline__ bool solver_helper_entry(const float *xbuf, int *hs) {
    if (blockDim.x <= 64) return false;
    if (threadIdx.x == 0) { hs[0] = 0; hs[1] = 0; }
    __syncthreads();
    if (threadIdx.x < 64) return false;
    norm_helper_wave(xbuf, hs);
    return true;
}
__device__ __forceinline__ void solver_helper_stop(int *hs) {
    if (blockDim.x > 64 && threadIdx.x == 0) hs_store(&hs[0], -1);
}

// (row sum - 1) / m of the rows of rowbuf in ATen's vectorised inner-sum order (see row_sums_wave_vs), returned wave-uniform.
// The aligned 8-lane group g of the wave takes row 8p + g in pass p; the passes (two for 9..16 rows) advance in LOCKSTEP --
// the section is one dependent chain per pass (loads -> ILP partials -> tail -> 8-lane sequential combine -> step), a
// dependent VALU op costs ~8 cycles while an independent one issues in ~2, so two interleaved chains cost what one does.
template <int MT, int VS>
__device__ __forceinline__ void row_steps_wave_vs(const float *rowbuf, int n, int m, float fm, float rcp_m, float (&tr)[MT]) {
    constexpr int NP = (MT + 7) / 8;
    const int lane = threadIdx.x & 63, l = lane & 7, g = lane >> 3;
    float v[NP][VS > 0 ? VS : 1];
    f32x4 ta[NP], tb[NP];
    const bool long_tail = (m & 7) > 4;                // wave-uniform: more than 4 trailing scalars
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int r = p * 8 + g;
        const float *x = rowbuf + (r < n ? r : n - 1) * kRowStride;
#pragma unroll
        for (int i = 0; i < VS; ++i) v[p][i] = x[8 * i + l];
        ta[p] = *reinterpret_cast<const f32x4 *>(x + 8 * VS);                    // zero from column m on
        tb[p] = *reinterpret_cast<const f32x4 *>(x + 8 * VS + 4);
    }
    float s[NP];
    if (VS == 0) {                                     // scalar_inner_sum (m < 8): ILP-4 over single elements
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
            if (m >= 4) {
                p0 = p0 + ta[p].x; p1 = p1 + ta[p].y; p2 = p2 + ta[p].z; p3 = p3 + ta[p].w;
                p0 = p0 + tb[p].x; p0 = p0 + tb[p].y; p0 = p0 + tb[p].z;
            } else {
                p0 = p0 + ta[p].x; p0 = p0 + ta[p].y; p0 = p0 + ta[p].z;
            }
            p0 = p0 + p1; p0 = p0 + p2; p0 = p0 + p3;
            s[p] = p0;
        }
    } else {
        constexpr int GQ = VS / 4;
        float p0[NP], p1[NP], p2[NP], p3[NP], a[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) { p0[p] = 0.0f; p1[p] = 0.0f; p2[p] = 0.0f; p3[p] = 0.0f; }
        if (GQ >= 1) {
#pragma unroll
            for (int p = 0; p < NP; ++p) { p0[p] = p0[p] + v[p][0]; p1[p] = p1[p] + v[p][1]; p2[p] = p2[p] + v[p][2]; p3[p] = p3[p] + v[p][3]; }
        }
        if (GQ >= 2) {
#pragma unroll
            for (int p = 0; p < NP; ++p) { p0[p] = p0[p] + v[p][4]; p1[p] = p1[p] + v[p][5]; p2[p] = p2[p] + v[p][6]; p3[p] = p3[p] + v[p][7]; }
        }
#pragma unroll
        for (int i = 4 * GQ; i < VS; ++i)
#pragma unroll
            for (int p = 0; p < NP; ++p) p0[p] = p0[p] + v[p][i];
#pragma unroll
        for (int p = 0; p < NP; ++p) p0[p] = p0[p] + p1[p];
#pragma unroll
        for (int p = 0; p < NP; ++p) p0[p] = p0[p] + p2[p];
#pragma unroll
        for (int p = 0; p < NP; ++p) p0[p] = p0[p] + p3[p];                      // vec[l]
        // trailing scalars, sequential from 0; adding the zeros past column m is exact, so with <= 4 of them the chain
        // stops after the fourth
#pragma unroll
        for (int p = 0; p < NP; ++p) a[p] = 0.0f + ta[p].x;
#pragma unroll
        for (int p = 0; p < NP; ++p) a[p] = a[p] + ta[p].y;
#pragma unroll
        for (int p = 0; p < NP; ++p) a[p] = a[p] + ta[p].z;
#pragma unroll
        for (int p = 0; p < NP; ++p) a[p] = a[p] + ta[p].w;
        if (long_tail) {
#pragma unroll
            for (int p = 0; p < NP; ++p) a[p] = a[p] + tb[p].x;
#pragma unroll
            for (int p = 0; p < NP; ++p) a[p] = a[p] + tb[p].y;
#pragma unroll
            for (int p = 0; p < NP; ++p) a[p] = a[p] + tb[p].z;
        }
        // acc + vec[0] + vec[1] + ... + vec[7] in that order (torder::add_group8_seq), the passes in lockstep
#pragma unroll
        for (int p = 0; p < NP; ++p) s[p] = a[p] + p0[p];
#pragma unroll
        for (int p = 0; p < NP; ++p) s[p] = s[p] + torder::shl_f32<1>(p0[p]);
#pragma unroll
        for (int p = 0; p < NP; ++p) s[p] = s[p] + torder::shl_f32<2>(p0[p]);
#pragma unroll
        for (int p = 0; p < NP; ++p) s[p] = s[p] + torder::shl_f32<3>(p0[p]);
#pragma unroll
        for (int p = 0; p < NP; ++p) s[p] = s[p] + torder::shl_f32<4>(p0[p]);
#pragma unroll
        for (int p = 0; p < NP; ++p) s[p] = s[p] + torder::shl_f32<5>(p0[p]);
#pragma unroll
        for (int p = 0; p < NP; ++p) s[p] = s[p] + torder::shl_f32<6>(p0[p]);
#pragma unroll
        for (int p = 0; p < NP; ++p) s[p] = s[p] + torder::shl_f32<7>(p0[p]);
    }
    // (row sum - 1) / m once per row, where the sum lives (div_by_const, the passes in lockstep)
    float q0[NP], rr[NP], step[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) s[p] = s[p] - 1.0f;
#pragma unroll
    for (int p = 0; p < NP; ++p) q0[p] = s[p] * rcp_m;
#pragma unroll
    for (int p = 0; p < NP; ++p) rr[p] = __builtin_fmaf(-q0[p], fm, s[p]);
#pragma unroll
    for (int p = 0; p < NP; ++p) step[p] = __builtin_fmaf(rr[p], rcp_m, q0[p]);
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (p * 8 + k < MT) tr[p * 8 + k] = readlane_f32(step[p], 8 * k);
}
// The same row steps for MORE THAN 8 ROWS with 4 lanes per row: lane q of an aligned 4-lane group carries ATen's vector
// lanes q and q + 4 of its row as the two halves of a packed fp32 pair, so 16 rows go through one pass -- for 9..16 rows
// one pass instead of two.  The one-wave sweep is ISSUE bound (~260 instructions x ~3.6 cycles, r04 ISA count in LABLOG),
// so the second pass was not free: this form issues ~30 instructions less per sweep.  Same adds in the same order:
// v_pk_add_f32 rounds its halves separately, and the in-order combine acc + vec[0] + ... + vec[7] walks the four low
// halves, then the four high halves, of the group.
template <int MT, int VS>
__device__ __forceinline__ void row_steps_wave_pk(const float *rowbuf, int n, int m, float fm, float rcp_m, float (&tr)[MT]) {
    constexpr int NP = (MT + 15) / 16;
    const int lane = threadIdx.x & 63, q = lane & 3, g = lane >> 2;
    f32x2 v[NP][VS > 0 ? VS : 1];
    f32x4 ta[NP], tb[NP];
    const bool long_tail = (m & 7) > 4;                // wave-uniform: more than 4 trailing scalars
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int r = p * 16 + g;
        const float *x = rowbuf + (r < n ? r : n - 1) * kRowStride;
#pragma unroll
        for (int i = 0; i < VS; ++i) v[p][i] = f32x2{x[8 * i + q], x[8 * i + q + 4]};
        ta[p] = *reinterpret_cast<const f32x4 *>(x + 8 * VS);                    // zero from column m on
        tb[p] = *reinterpret_cast<const f32x4 *>(x + 8 * VS + 4);
    }
    static_assert(VS >= 1, "rows shorter than 8 columns take the scalar form");
    constexpr int GQ = VS / 4;
    const f32x2 z2 = f32x2{0.0f, 0.0f};
    f32x2 p0[NP], p1[NP], p2[NP], p3[NP];
    float a[NP], s[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) { p0[p] = z2; p1[p] = z2; p2[p] = z2; p3[p] = z2; }
    if (GQ >= 1) {
#pragma unroll
        for (int p = 0; p < NP; ++p) { p0[p] = p0[p] + v[p][0]; p1[p] = p1[p] + v[p][1]; p2[p] = p2[p] + v[p][2]; p3[p] = p3[p] + v[p][3]; }
    }
    if (GQ >= 2) {
#pragma unroll
        for (int p = 0; p < NP; ++p) { p0[p] = p0[p] + v[p][4]; p1[p] = p1[p] + v[p][5]; p2[p] = p2[p] + v[p][6]; p3[p] = p3[p] + v[p][7]; }
    }
#pragma unroll
    for (int i = 4 * GQ; i < VS; ++i)
#pragma unroll
        for (int p = 0; p < NP; ++p) p0[p] = p0[p] + v[p][i];
#pragma unroll
    for (int p = 0; p < NP; ++p) p0[p] = p0[p] + p1[p];
#pragma unroll
    for (int p = 0; p < NP; ++p) p0[p] = p0[p] + p2[p];
#pragma unroll
    for (int p = 0; p < NP; ++p) p0[p] = p0[p] + p3[p];                          // {vec[q], vec[q + 4]}
#pragma unroll
    for (int p = 0; p < NP; ++p) a[p] = 0.0f + ta[p].x;
#pragma unroll
    for (int p = 0; p < NP; ++p) a[p] = a[p] + ta[p].y;
#pragma unroll
    for (int p = 0; p < NP; ++p) a[p] = a[p] + ta[p].z;
#pragma unroll
    for (int p = 0; p < NP; ++p) a[p] = a[p] + ta[p].w;
    if (long_tail) {
#pragma unroll
        for (int p = 0; p < NP; ++p) a[p] = a[p] + tb[p].x;
#pragma unroll
        for (int p = 0; p < NP; ++p) a[p] = a[p] + tb[p].y;
#pragma unroll
        for (int p = 0; p < NP; ++p) a[p] = a[p] + tb[p].z;
    }
    // acc + vec[0] + vec[1] + ... + vec[7] in that order, valid in lane 0 of every 4-lane group
#pragma unroll
    for (int p = 0; p < NP; ++p) s[p] = a[p] + p0[p].x;
#pragma unroll
    for (int p = 0; p < NP; ++p) s[p] = s[p] + torder::shl_f32<1>(p0[p].x);
#pragma unroll
    for (int p = 0; p < NP; ++p) s[p] = s[p] + torder::shl_f32<2>(p0[p].x);
#pragma unroll
    for (int p = 0; p < NP; ++p) s[p] = s[p] + torder::shl_f32<3>(p0[p].x);
#pragma unroll
    for (int p = 0; p < NP; ++p) s[p] = s[p] + p0[p].y;
#pragma unroll
    for (int p = 0; p < NP; ++p) s[p] = s[p] + torder::shl_f32<1>(p0[p].y);
#pragma unroll
    for (int p = 0; p < NP; ++p) s[p] = s[p] + torder::shl_f32<2>(p0[p].y);
#pragma unroll
    for (int p = 0; p < NP; ++p) s[p] = s[p] + torder::shl_f32<3>(p0[p].y);
    float q0[NP], rr[NP], step[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) s[p] = s[p] - 1.0f;
#pragma unroll
    for (int p = 0; p < NP; ++p) q0[p] = s[p] * rcp_m;
#pragma unroll
    for (int p = 0; p < NP; ++p) rr[p] = __builtin_fmaf(-q0[p], fm, s[p]);
#pragma unroll
    for (int p = 0; p < NP; ++p) step[p] = __builtin_fmaf(rr[p], rcp_m, q0[p]);
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (p * 16 + k < MT) tr[p * 16 + k] = readlane_f32(step[p], 4 * k);
}

// VSC >= 0: m >> 3 == VSC is known at compile time (relax_core_w1 is compiled once more for the model's width); -1: any m.
template <int MT, int VSC = -1>
__device__ __forceinline__ void row_steps_wave(const float *rowbuf, int n, int m, float fm, float rcp_m, float (&tr)[MT]) {
    __builtin_amdgcn_wave_barrier();                   // scheduling fence only: the wave's own LDS writes are ordered
    if constexpr (VSC >= 1) {
        if constexpr (MT > 8) row_steps_wave_pk<MT, VSC>(rowbuf, n, m, fm, rcp_m, tr);
        else row_steps_wave_vs<MT, VSC>(rowbuf, n, m, fm, rcp_m, tr);
        return;
    }
    if constexpr (MT > 8) {
        switch (m >> 3) {
            case 0: row_steps_wave_vs<MT, 0>(rowbuf, n, m, fm, rcp_m, tr); break;
            case 1: row_steps_wave_pk<MT, 1>(rowbuf, n, m, fm, rcp_m, tr); break;
            case 2: row_steps_wave_pk<MT, 2>(rowbuf, n, m, fm, rcp_m, tr); break;
            case 3: row_steps_wave_pk<MT, 3>(rowbuf, n, m, fm, rcp_m, tr); break;
            case 4: row_steps_wave_pk<MT, 4>(rowbuf, n, m, fm, rcp_m, tr); break;
            case 5: row_steps_wave_pk<MT, 5>(rowbuf, n, m, fm, rcp_m, tr); break;
            case 6: row_steps_wave_pk<MT, 6>(rowbuf, n, m, fm, rcp_m, tr); break;
            case 7: row_steps_wave_pk<MT, 7>(rowbuf, n, m, fm, rcp_m, tr); break;
            default: row_steps_wave_pk<MT, 8>(rowbuf, n, m, fm, rcp_m, tr); break;
        }
        return;
    }
    switch (m >> 3) {
        case 0: row_steps_wave_vs<MT, 0>(rowbuf, n, m, fm, rcp_m, tr); break;
        case 1: row_steps_wave_vs<MT, 1>(rowbuf, n, m, fm, rcp_m, tr); break;
        case 2: row_steps_wave_vs<MT, 2>(rowbuf, n, m, fm, rcp_m, tr); break;
        case 3: row_steps_wave_vs<MT, 3>(rowbuf, n, m, fm, rcp_m, tr); break;
        case 4: row_steps_wave_vs<MT, 4>(rowbuf, n, m, fm, rcp_m, tr); break;
        case 5: row_steps_wave_vs<MT, 5>(rowbuf, n, m, fm, rcp_m, tr); break;
        case 6: row_steps_wave_vs<MT, 6>(rowbuf, n, m, fm, rcp_m, tr); break;
        case 7: row_steps_wave_vs<MT, 7>(rowbuf, n, m, fm, rcp_m, tr); break;
        default: row_steps_wave_vs<MT, 8>(rowbuf, n, m, fm, rcp_m, tr); break;
    }
}


// (Tried and dropped, round 3: the same row steps WITHOUT the LDS round trip -- ATen's 8-wide vector i of a row is the lane
// group [8i, 8i + 8) of the row's register, so the ILP partials can be added in registers with row_shl:8 and gfx950's
// v_permlane16_swap / v_permlane32_swap, eight rows transposed into the eight lane groups of one register, one DPP chain
// for all of them.  Bit identical through every solver golden, but SLOWER: 10 x 50: 107.5 vs 72.6 us, 5 x 50 at 40 x 5:
// 145.5 vs 99.3 us -- the permlane swaps cost far more than the LDS turn-around they replace.)

// VSC: see row_steps_wave.  The switch over the width class inside the sweep cost ~10 % of it (the compare tree's taken
// branches are instruction-fetch bubbles for a wave alone on its SIMD, and the row sums could not be scheduled into the
// element-wise work around them): 10 x 50 at 20 x 5 64.5 -> 58.3 us, 5 x 50 50.5 -> 44.7 us with the width class fixed.
// So the core is compiled for the model's width classes (4, 5, 6: 32..55 columns -- the evaluator keeps up to 50 proposals
// per frame, 30..50 after NMS) and once more for any width.
// Tape (for the backward): per executed sweep and thread one uint2 {relu bits (bit i <=> row i passed the relu),
// column-over flag}; per outer iteration the number of executed sweeps.
struct RelaxTape {
    uint2 *bits;        // global [max_iter * proj_iter][64 * NG]
    int *sweeps;        // [max_iter] (LDS in the backward's re-run)
};

template <int MT, bool EXACT, int VSC = -1, bool TAPE = false>
__device__ __forceinline__ int relax_core_w1(const float (&C)[MT], int n_rt, int m, int col, const RelaxParams prm,
                                             float *xbuf, float *rsbuf, int *hs, float (&X)[MT], float (&acc)[MT],
                                             float *cost_out, float *rowbuf /* LDS, MT * kRowStride + 8 floats, 16-byte aligned */,
                                             RelaxTape tape = RelaxTape{nullptr, nullptr}) {
    constexpr int MP = (MT + 1) / 2;                   // row pairs; an odd MT leaves a dummy slot that stays zero
    int tape_pos = 0;
    const int n = EXACT ? MT : n_rt;
    const bool with_helper = blockDim.x > 64;          // wave 1 computes the cost norms (norm_helper_wave)
    if (with_helper && threadIdx.x == 0) hs[3] = n * m;
    if (threadIdx.x < 8) {
#pragma unroll
        for (int i = 0; i < MT; ++i) rowbuf[i * kRowStride + 64 + threadIdx.x] = 0.0f;
        rowbuf[MT * kRowStride + threadIdx.x] = 0.0f;
    }
#define DMM_ROW(i) (EXACT || (i) < n)
    const bool live = col < m;
    const float fn = (float)n, fm = (float)m;
    const float rcp_n = 1.0f / fn, rcp_m = 1.0f / fm;
    const bool col_class_a = col < torder::outer_class_bound(m);
    const int n4 = 4 * (n / 4);

    // ---- greedy row-min initialisation (relax_match.py:45-55); max / first-argmin are order free ----
    float cmax = -__builtin_inff();
#pragma unroll
    for (int i = 0; i < MT; ++i)
        if (DMM_ROW(i) && live) cmax = C[i] > cmax ? C[i] : cmax;
    cmax = wave_max(cmax);
    int best_row = 0;
    {
        float bv = C[0];
#pragma unroll
        for (int i = 1; i < MT; ++i)
            if (DMM_ROW(i) && C[i] < bv) { bv = C[i]; best_row = i; }   // first argmin over rows
    }
    {
        float crm[MT], vmin[MT];
        int cand[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            crm[i] = (live && DMM_ROW(i)) ? (i == best_row ? C[i] : cmax) : __builtin_inff();
            vmin[i] = crm[i];
        }
        wave_min_rows<MT>(vmin);
#pragma unroll
        for (int i = 0; i < MT; ++i) cand[i] = (live && crm[i] == vmin[i]) ? col : 0x7fffffff;
        wave_min_rows_i32<MT>(cand);                                 // first argmin over columns
#pragma unroll
        for (int i = 0; i < MT; ++i) X[i] = (DMM_ROW(i) && col == cand[i]) ? 1.0f : 0.0f;
    }
    f32x2 Xp[MP], Cp[MP], P0[MP], P1[MP], P2[MP], ap[MP];
#pragma unroll
    for (int k = 0; k < MP; ++k) {
        Xp[k] = f32x2{X[2 * k], 2 * k + 1 < MT ? X[2 * k + 1] : 0.0f};
        Cp[k] = f32x2{C[2 * k], 2 * k + 1 < MT ? C[2 * k + 1] : 0.0f};
        P0[k] = f32x2{0.0f, 0.0f}; P1[k] = P0[k]; P2[k] = P0[k];
        ap[k] = f32x2{0.0f, 0.0f} + Xp[k];                        // sum(X_list) starts at 0 + X0
    }
    if (cost_out && threadIdx.x == 0) cost_out[0] = 0.0f;
    const f32x2 lr2 = f32x2{prm.lr, prm.lr};
    constexpr float kMoveThr = 0x1p-75f;               // d*d != 0 in fp32  <=>  !(|d| <= 2^-75)

    int len = 1;
    float cost_prev = 0.0f;
    for (int it = 0; it < prm.max_iter; ++it) {
        // gradient step X = X - lr*C (:69); cost = ||X*C||_F (:70); X_list.append(X) (:71)
#pragma unroll
        for (int k = 0; k < MP; ++k) {
            const f32x2 g = lr2 * Cp[k];
            Xp[k] = Xp[k] - g;
            ap[k] = ap[k] + Xp[k];
        }
        if (live) {
#pragma unroll
            for (int k = 0; k < MP; ++k) {
                const f32x2 pr = Xp[k] * Cp[k];
                xbuf[(2 * k) * m + col] = pr.x;                      // rows >= n land past the n x m matrix
                if (2 * k + 1 < MT) xbuf[(2 * k + 1) * m + col] = pr.y;
            }
        }
        float cost = 0.0f;
        if (with_helper) {
            if (threadIdx.x == 0) hs_store(&hs[0], it + 1);     // after this wave's product writes (LDS is in order)
        } else {
            cost = norm_torch_order(xbuf, n * m, rsbuf + MT);
        }
        ++len;

        // `if ||X - X_start|| == 0: break` (:88-89) is decided ONE PHASE LATE: the lane masks of sweep j are tested after
        // the relu step and the column sums of sweep j + 1 have been issued (a branch right behind the compares stalled
        // the wave for the whole compare -> scalar -> branch latency, ~0.1 us per sweep); when sweep j turns out to have
        // moved nothing, the relu step of sweep j + 1 -- all that was done since: it only touches X and P0 -- is undone.
        unsigned long long moved_prev = ~0ull;
        int sweeps_done = 0;
        // one sweep; true = the sweep before it moved nothing (stop).  Called twice per trip of the loop below: X and
        // X_start trade registers from one sweep to the next, and a rolled loop paid for that with 20 v_mov per sweep.
        auto sweep = [&]() -> bool {
            f32x2 Xs[MP], P0s[MP];
            unsigned relu_bits = 0;
            // {X >= 0} (:74-76) then X = Y + P1 (:78)
#pragma unroll
            for (int k = 0; k < MP; ++k) {
                Xs[k] = Xp[k];
                P0s[k] = P0[k];
                const f32x2 x = Xp[k] + P0[k];
                const f32x2 y = f32x2{__builtin_fmaxf(x.x, 0.0f), __builtin_fmaxf(x.y, 0.0f)};
                if (TAPE) relu_bits |= (x.x > 0.0f ? 1u << (2 * k) : 0u) | (x.y > 0.0f ? 2u << (2 * k) : 0u);
                P0[k] = x - y;
                Xp[k] = y + P1[k];
            }
            // X.sum(dim=0) in ATen's outer-sum order for this column's class (in-lane)
            float cs;
            {
                float a0 = 0.0f, a1 = 0.0f;                 // class A: one cascade chain, 16-row blocks
                float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;   // class B: ILP-4 row_sum
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const float xi = (i & 1) ? Xp[i >> 1].y : Xp[i >> 1].x;
                    a0 = a0 + xi;
                    if (((i + 1) & 15) == 0) { a1 = a1 + a0; a0 = 0.0f; }
                    const float xm = (EXACT ? i < 4 * (MT / 4) : i < n4) ? xi : 0.0f;
                    if ((i & 3) == 0) p0 = p0 + xm;
                    else if ((i & 3) == 1) p1 = p1 + xm;
                    else if ((i & 3) == 2) p2 = p2 + xm;
                    else p3 = p3 + xm;
                }
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const float xi = (i & 1) ? Xp[i >> 1].y : Xp[i >> 1].x;
                    if (EXACT) { if (i >= 4 * (MT / 4)) p0 = p0 + xi; }
                    else p0 = p0 + (i >= n4 ? xi : 0.0f);
                }
                p0 = p0 + p1;
                p0 = p0 + p2;
                p0 = p0 + p3;
                cs = col_class_a ? a0 + a1 : p0;
            }
            if (moved_prev == 0ull) {                          // sweep j - 1 was the last one (:88-89)
#pragma unroll
                for (int k = 0; k < MP; ++k) { Xp[k] = Xs[k]; P0[k] = P0s[k]; }
                return true;
            }
            // {column sums <= 1}: project_col (:21-34, :79-80); then X = Y + P2 (:82)
            const bool over = cs > 1.0f;                       // mask = (X_col_sum <= 1)
            if (TAPE) {                                        // (a sweep undone above was never taped)
                tape.bits[(size_t)tape_pos * 64 + threadIdx.x] = make_uint2(relu_bits, over ? 1u : 0u);
                ++tape_pos;
                ++sweeps_done;
            }
            float tc = div_by_const(cs - 1.0f, fn, rcp_n);
            tc = over ? tc : 0.0f;                             // x - 0 = x: the reference's `Y = X` branch, exactly
#pragma unroll
            for (int k = 0; k < MP; ++k) {
                const f32x2 tcp = f32x2{DMM_ROW(2 * k) ? tc : 0.0f, (2 * k + 1 < MT && DMM_ROW(2 * k + 1)) ? tc : 0.0f};
                const f32x2 x = Xp[k];
                const f32x2 y = x - tcp;
                P1[k] = x - y;
                Xp[k] = y + P2[k];
            }
            // {row sums = 1}: project_row (:9-19, :83-84); X.sum(dim=1) in ATen's inner-sum order
            float tr[MT];
#pragma unroll
            for (int k = 0; k < MP; ++k) {                     // every lane: dead columns hold exact zeros
                rowbuf[(2 * k) * kRowStride + col] = Xp[k].x;
                if (2 * k + 1 < MT) rowbuf[(2 * k + 1) * kRowStride + col] = Xp[k].y;
            }
            row_steps_wave<MT, VSC>(rowbuf, n, m, fm, rcp_m, tr);
            bool moved = false;
            if (live) {                                        // dead columns keep their zeros
#pragma unroll
                for (int k = 0; k < MP; ++k) {
                    const f32x2 trp = f32x2{DMM_ROW(2 * k) ? tr[2 * k] : 0.0f,
                                            (2 * k + 1 < MT && DMM_ROW(2 * k + 1)) ? tr[2 * k + 1 < MT ? 2 * k + 1 : 0] : 0.0f};
                    const f32x2 x = Xp[k];
                    const f32x2 y = x - trp;
                    P2[k] = x - y;
                    Xp[k] = y;                                  // :86
                    const f32x2 d = y - Xs[k];
                    moved |= !(__builtin_fabsf(d.x) <= kMoveThr);
                    moved |= !(__builtin_fabsf(d.y) <= kMoveThr);
                }
            }
            // a sum of squares is zero iff every square rounds to zero: "no lane saw a move" is the reference's test
            moved_prev = __ballot(moved);
            return false;
        };
        if (prm.proj_iter == 5) {                             // the reference's setting (relax_proj_iter: 5), straight line
            if (!sweep() && !sweep() && !sweep() && !sweep()) (void)sweep();
        } else {
            for (int j = 0; j < prm.proj_iter; j += 2) {
                if (sweep()) break;
                if (j + 1 >= prm.proj_iter) break;
                if (sweep()) break;
            }
        }
        if (with_helper) {
            while (hs_load(&hs[1]) != it + 1) {}
            cost = __int_as_float(hs[2]);
        }
        if (TAPE && threadIdx.x == 0) tape.sweeps[it] = sweeps_done;
        if (cost_out && threadIdx.x == 0) cost_out[it + 1] = cost;
        if (cost_prev == cost) break;                           // :96-98
        cost_prev = cost;
    }
    solver_helper_stop(hs);
#undef DMM_ROW
#pragma unroll
    for (int k = 0; k < MP; ++k) {
        X[2 * k] = Xp[k].x;
        acc[2 * k] = ap[k].x;
        if (2 * k + 1 < MT) { X[2 * k + 1] = Xp[k].y; acc[2 * k + 1] = ap[k].y; }
    }
    return len - 1;
}

// ---------------------------------------------------------------------------------------------
// relax_matching core.  C[i] = cost of (row i, this thread's column); n rows, m columns live.
// Threads with col >= m carry zeros everywhere and never change.  On return X[] is the final
// projected iterate, acc[] = sum(X_list); returns len(X_list) - 1.
// EXACT: n == MT at compile time (all row guards fold away).
// xbuf: LDS [MT * 64 * NG] floats, rsbuf: LDS [MT + 1] floats.
// ---------------------------------------------------------------------------------------------
template <int MT, int NG, bool EXACT, bool TAPE = false, bool W1TAPE = false>
__device__ __forceinline__ int relax_core(const float (&C)[MT], int n_rt, int m, int col, const RelaxParams prm,
                                          BlockRed<MT, NG> &red, float *xbuf, float *rsbuf, float (&X)[MT],
                                          float (&acc)[MT], float *cost_out /* global [max_iter+1] or null */,
                                          RelaxTape tape = RelaxTape{nullptr, nullptr}, int *hs = nullptr) {
#ifndef DMM_SOLVER_NO_W1
    if constexpr (NG == 1 && !TAPE) {                  // forward, one wave per frame: the latency form
        (void)red;
        __shared__ __attribute__((aligned(16))) float rowbuf_w1[MT * kRowStride + 8];   // one buffer for all the variants
        // W1TAPE (the kernels of dmm_match_train_forward; kernels of their own: folded into the untaped ones as a run-time
        // choice the evaluator's solve measured 0.9 us slower, 78.5 vs 77.6 us at 5 x 50, 40 x 5): the training forward tapes
        // its sweeps for the backward, which then skips its re-run of the solver.  (TAPE = true -- the backward's own re-run
        // when no tape was kept -- stays on the form below: without the helper wave it is the faster one, 48.3 vs 50.9 us)
        if constexpr (W1TAPE) {
            if ((m >> 3) == 6)                         // 48..55 columns: the model's 50 proposals
                return relax_core_w1<MT, EXACT, 6, true>(C, n_rt, m, col, prm, xbuf, rsbuf, hs, X, acc, cost_out, rowbuf_w1, tape);
            return relax_core_w1<MT, EXACT, -1, true>(C, n_rt, m, col, prm, xbuf, rsbuf, hs, X, acc, cost_out, rowbuf_w1, tape);
        }
        switch (m >> 3) {                              // the evaluator's frames: 50 proposals, 32..55 columns after NMS
            case 4: return relax_core_w1<MT, EXACT, 4>(C, n_rt, m, col, prm, xbuf, rsbuf, hs, X, acc, cost_out, rowbuf_w1);
            case 5: return relax_core_w1<MT, EXACT, 5>(C, n_rt, m, col, prm, xbuf, rsbuf, hs, X, acc, cost_out, rowbuf_w1);
            case 6: return relax_core_w1<MT, EXACT, 6>(C, n_rt, m, col, prm, xbuf, rsbuf, hs, X, acc, cost_out, rowbuf_w1);
            default: return relax_core_w1<MT, EXACT>(C, n_rt, m, col, prm, xbuf, rsbuf, hs, X, acc, cost_out, rowbuf_w1);
        }
    }
#endif
    int tape_pos = 0;
    const int n = EXACT ? MT : n_rt;
    // NG == 1: the zero-padded row buffer of row_sums_torch_order_wave
    __shared__ float rowbuf[NG == 1 ? MT * kRowStride : 1];
    if (NG == 1 && threadIdx.x < 8) {
#pragma unroll
        for (int i = 0; i < MT; ++i) rowbuf[i * kRowStride + 64 + threadIdx.x] = 0.0f;
    }
#define DMM_ROW(i) (EXACT || (i) < n)
    const bool live = col < m;
    const float fn = (float)n, fm = (float)m;
    const float rcp_n = 1.0f / fn, rcp_m = 1.0f / fm;
    const bool col_class_a = col < torder::outer_class_bound(m);   // ATen outer-sum class of this column
    const int n4 = 4 * (n / 4);

    // ---- greedy row-min initialisation (relax_match.py:45-55); max / first-argmin are order free ----
    float cm[1] = {-__builtin_inff()};
#pragma unroll
    for (int i = 0; i < MT; ++i)
        if (DMM_ROW(i) && live) cm[0] = C[i] > cm[0] ? C[i] : cm[0];
    cm[0] = wave_max(cm[0]);
    red.fold(cm, fmax_op());
    const float cmax = cm[0];
    int best_row = 0;
    {
        float bv = C[0];
#pragma unroll
        for (int i = 1; i < MT; ++i)
            if (DMM_ROW(i) && C[i] < bv) { bv = C[i]; best_row = i; }   // first argmin over rows
    }
    {
        float crm[MT], vmin[MT];
        int cand[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            // C_rowmin[i, col]; dead columns / rows are +inf so they never win the row argmin
            crm[i] = (live && DMM_ROW(i)) ? (i == best_row ? C[i] : cmax) : __builtin_inff();
            vmin[i] = crm[i];
        }
        wave_min_rows<MT>(vmin);
        red.fold(vmin, fmin_op());
#pragma unroll
        for (int i = 0; i < MT; ++i) cand[i] = (live && crm[i] == vmin[i]) ? col : 0x7fffffff;
        wave_min_rows_i32<MT>(cand);
        red.min_i32(cand);                                           // first argmin over columns
#pragma unroll
        for (int i = 0; i < MT; ++i) X[i] = (DMM_ROW(i) && col == cand[i]) ? 1.0f : 0.0f;
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
#pragma unroll
        for (int i = 0; i < MT; ++i) {                           // rows >= n hold zeros in C, X, P*: no guards needed
            const float g = prm.lr * C[i];
            X[i] = X[i] - g;
            acc[i] = acc[i] + X[i];
        }
        if (live) {                                              // one predicated block, not one exec dance per row
#pragma unroll
            for (int i = 0; i < MT; ++i) xbuf[i * m + col] = X[i] * C[i];   // rows >= n land past the n x m matrix
        }
        const float cost = norm_torch_order(xbuf, n * m, rsbuf + MT);
        if (cost_out && threadIdx.x == 0) cost_out[it + 1] = cost;
        ++len;

        int sweeps_done = 0;
        for (int j = 0; j < prm.proj_iter; ++j) {
            float Xs[MT];
            unsigned relu_bits = 0;
            // {X >= 0} (:74-76) then X = Y + P1 (:78)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                Xs[i] = X[i];
                float x = X[i] + P0[i];
                const float y = x > 0.0f ? x : 0.0f;
                if (TAPE && x > 0.0f) relu_bits |= 1u << i;
                P0[i] = x - y;
                X[i] = y + P1[i];
            }
            // X.sum(dim=0) in ATen's outer-sum order for this column's class (in-lane)
            float cs;
            {
                float a0 = 0.0f, a1 = 0.0f;                 // class A: one cascade chain, 16-row blocks
                float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;   // class B: ILP-4 row_sum
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    // adding the zeros of rows >= n (and +0.0f for the rows a chain does not own) is exact, so the
                    // row guards become selects on wave-uniform conditions instead of branches
                    a0 = a0 + X[i];
                    if (((i + 1) & 15) == 0) { a1 = a1 + a0; a0 = 0.0f; }
                    const float xm = (EXACT ? i < 4 * (MT / 4) : i < n4) ? X[i] : 0.0f;
                    if ((i & 3) == 0) p0 = p0 + xm;
                    else if ((i & 3) == 1) p1 = p1 + xm;
                    else if ((i & 3) == 2) p2 = p2 + xm;
                    else p3 = p3 + xm;
                }
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    if (EXACT) { if (i >= 4 * (MT / 4)) p0 = p0 + X[i]; }
                    else p0 = p0 + (i >= n4 ? X[i] : 0.0f);
                }
                p0 = p0 + p1;
                p0 = p0 + p2;
                p0 = p0 + p3;
                cs = col_class_a ? a0 + a1 : p0;
            }
            // {column sums <= 1}: project_col (:21-34, :79-80); then X = Y + P2 (:82)
            const bool over = cs > 1.0f;                       // mask = (X_col_sum <= 1)
            if (TAPE) {
                tape.bits[(size_t)tape_pos * (64 * NG) + threadIdx.x] = make_uint2(relu_bits, over ? 1u : 0u);
                ++tape_pos;
            }
            ++sweeps_done;
            const float tc = div_by_const(cs - 1.0f, fn, rcp_n);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                float x = X[i];
                const float tci = DMM_ROW(i) ? tc : 0.0f;        // rows >= n stay zero
                const float y = over ? x - tci : x;
                P1[i] = x - y;
                x = y + P2[i];
                X[i] = x;
            }
            if (NG == 1) {                                        // every lane: dead columns hold exact zeros
#pragma unroll
                for (int i = 0; i < MT; ++i) rowbuf[i * kRowStride + col] = X[i];
            } else if (live) {
#pragma unroll
                for (int i = 0; i < MT; ++i) xbuf[i * m + col] = X[i];
            }
            // {row sums = 1}: project_row (:9-19, :83-84); X.sum(dim=1) in ATen's inner-sum order
            float rsv[MT], trv[MT];
            if (NG == 1) {
                row_sums_torch_order_wave<MT>(rowbuf, n, m, rsv);
#pragma unroll
                for (int i = 0; i < MT; ++i) trv[i] = div_by_const(rsv[i] - 1.0f, fm, rcp_m);
            } else {
                row_steps_torch_order<NG>(xbuf, n, m, fm, rcp_m, rsbuf);
#pragma unroll
                for (int i = 0; i < MT; ++i) trv[i] = DMM_ROW(i) ? rsbuf[i] : 0.0f;
            }
            unsigned moved_bits = 0;                            // OR of the squares' bit patterns: non-zero <=> some square
#pragma unroll                                                  // is non-zero (a NaN has non-zero bits: "moved")
            for (int i = 0; i < MT; ++i) {
                float tr = trv[i];
                tr = (live && DMM_ROW(i)) ? tr : 0.0f;          // dead columns / rows keep their zeros (x - 0 = x, P2 = 0)
                const float x = X[i];
                const float y = x - tr;
                P2[i] = x - y;
                X[i] = y;                                       // :86
                const float d = y - Xs[i];
                const float sq = d * d;
                moved_bits |= __float_as_uint(sq);
            }
            const bool moved = moved_bits != 0u;
            // if ||X - X_start|| == 0: break (:88-89).  A sum of squares is zero iff every square rounds to zero,
            // whatever the order: "no lane saw a non-zero square" is exactly the reference's decision.
            float mv[1] = {__ballot(moved) != 0ull ? 1.0f : 0.0f};
            red.fold(mv, fmax_op());
            if (mv[0] == 0.0f) break;
        }
        if (TAPE && threadIdx.x == 0) tape.sweeps[it] = sweeps_done;
        if (cost_prev == cost) break;                           // :96-98
        cost_prev = cost;
    }
#undef DMM_ROW
    return len - 1;
}


// ---------------------------------------------------------------------------------------------
// relax_matching core with the solver STATE IN PACKED FP16 and fp32 sums (BASELINE configs[4]: "fp16 Sinkhorn with fp32
// accumulate"); opt-in (dmm_relax_match_f16s), NOT bit exact -- a tolerance mode.  Same algorithm and control flow as
// relax_core (relax_match.py:36-105): projected gradient steps, Dykstra sweeps over {X >= 0}, {column sums <= 1},
// {row sums = 1}, both data-dependent exits, R = mean of the pre-projection iterates.  What changes:
//   * X, the three Dykstra increments and the sweep's start copy are half2 PAIRS OF ROWS per thread (5 registers per two
//     rows instead of 10): the 20 x 200 problem of config 5 needs ~110 VGPRs instead of 256 + spills, so four waves fit a
//     SIMD and the solver can run BESIDE the streaming count kernel (the fp32 form holds half of a CU's registers for
//     0.3 ms and the 2-lane schedule lost to the single stream);
//   * element-wise steps are v_pk_add_f16 / v_pk_max_f16 (two rows per instruction), the column sums v_dot2_f32_f16
//     against (1, 1) -- fp32 accumulation, two rows per instruction --, row sums / cost norm / sum of iterates are fp32
//     (DPP tree + LDS fold: no summation order to reproduce here);
//   * the exits compare the fp16 iterate bit for bit and the fp32 cost for equality: they fire when the fp16 iteration
//     has reached its fixed point, which need not be the step at which the fp32 reference's does.
// Tolerance (tests/test_gpu_parity.py): |R - R_fp32| <= 1e-2 where both ran the same number of iterations, identical
// row argmax wherever the fp32 decision is not a near tie.
// ---------------------------------------------------------------------------------------------
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

template <int MT, int NG>
__device__ __forceinline__ int relax_core_h(const float (&C)[MT], int n, int m, int col, const RelaxParams prm,
                                            BlockRed<MT, NG> &red, float *accbuf /* LDS [MT][64 NG] */, float (&X)[MT]) {
    constexpr int MP = (MT + 1) / 2;
    constexpr int LD = 64 * NG;
    float *acc_t = accbuf + threadIdx.x;               // sum(X_list) of this thread's column: LDS, touched once per
                                                       // outer iteration (20 registers less in the sweep)
#define DMM_ROWH(i) ((i) < n)
    const bool live = col < m;
    const float fn = (float)n, fm = (float)m;
    // ---- greedy row-min initialisation in fp32 (relax_match.py:45-55), as relax_core ----
    float cm[1] = {-__builtin_inff()};
#pragma unroll
    for (int i = 0; i < MT; ++i)
        if (DMM_ROWH(i) && live) cm[0] = C[i] > cm[0] ? C[i] : cm[0];
    cm[0] = wave_max(cm[0]);
    red.fold(cm, fmax_op());
    const float cmax = cm[0];
    int best_row = 0;
    {
        float bv = C[0];
#pragma unroll
        for (int i = 1; i < MT; ++i)
            if (DMM_ROWH(i) && C[i] < bv) { bv = C[i]; best_row = i; }
    }
    {
        float crm[MT], vmin[MT];
        int cand[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            crm[i] = (live && DMM_ROWH(i)) ? (i == best_row ? C[i] : cmax) : __builtin_inff();
            vmin[i] = crm[i];
        }
        wave_min_rows<MT>(vmin);
        red.fold(vmin, fmin_op());
#pragma unroll
        for (int i = 0; i < MT; ++i) cand[i] = (live && crm[i] == vmin[i]) ? col : 0x7fffffff;
        wave_min_rows_i32<MT>(cand);
        red.min_i32(cand);
#pragma unroll
        for (int i = 0; i < MT; ++i) X[i] = (DMM_ROWH(i) && col == cand[i]) ? 1.0f : 0.0f;
    }
    const h16x2 zero2 = {(_Float16)0.0f, (_Float16)0.0f}, one2 = {(_Float16)1.0f, (_Float16)1.0f};
    h16x2 Xh[MP], Ch[MP], P0[MP], P1[MP], P2[MP];
#pragma unroll
    for (int k = 0; k < MP; ++k) {
        Xh[k] = h16x2{(_Float16)X[2 * k], (_Float16)(2 * k + 1 < MT ? X[2 * k + 1] : 0.0f)};
        Ch[k] = h16x2{(_Float16)C[2 * k], (_Float16)(2 * k + 1 < MT ? C[2 * k + 1] : 0.0f)};
        P0[k] = zero2; P1[k] = zero2; P2[k] = zero2;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) acc_t[i * LD] = 0.0f + X[i];
    const h16x2 lr2 = {(_Float16)prm.lr, (_Float16)prm.lr};
    int len = 1;
    float cost_prev = 0.0f;
    for (int it = 0; it < prm.max_iter; ++it) {
        float cq[1] = {0.0f};
#pragma unroll
        for (int k = 0; k < MP; ++k) {
            Xh[k] = Xh[k] - lr2 * Ch[k];                           // X = X - lr*C (:69)
            acc_t[(2 * k) * LD] = acc_t[(2 * k) * LD] + (float)Xh[k].x;   // sum(X_list) in fp32
            if (2 * k + 1 < MT) acc_t[(2 * k + 1) * LD] = acc_t[(2 * k + 1) * LD] + (float)Xh[k].y;
            const h16x2 pr = Xh[k] * Ch[k];
            cq[0] = __builtin_amdgcn_fdot2(pr, pr, cq[0], false);  // ||X*C||_F^2, fp32 accumulate
        }
        cq[0] = wave_sum(cq[0]);
        red.sum(cq);
        const float cost = __builtin_sqrtf(cq[0]);
        ++len;
        for (int j = 0; j < prm.proj_iter; ++j) {
            h16x2 Xs[MP];
            float cs = 0.0f;
#pragma unroll
            for (int k = 0; k < MP; ++k) {
                Xs[k] = Xh[k];
                const h16x2 x = Xh[k] + P0[k];
                const h16x2 y = __builtin_elementwise_max(x, zero2);   // {X >= 0} (:74-76)
                P0[k] = x - y;
                Xh[k] = y + P1[k];                                 // (:78)
                cs = __builtin_amdgcn_fdot2(Xh[k], one2, cs, false);   // X.sum(dim=0), fp32
            }
            const bool over = cs > 1.0f;
            const _Float16 tch = (_Float16)(over ? (cs - 1.0f) / fn : 0.0f);
            float rs[MT];
#pragma unroll
            for (int k = 0; k < MP; ++k) {
                const h16x2 tcp = {DMM_ROWH(2 * k) ? tch : (_Float16)0.0f, DMM_ROWH(2 * k + 1) ? tch : (_Float16)0.0f};
                const h16x2 x = Xh[k];
                const h16x2 y = x - tcp;                           // project_col (:21-34)
                P1[k] = x - y;
                Xh[k] = y + P2[k];                                 // (:82)
                rs[2 * k] = (float)Xh[k].x;
                if (2 * k + 1 < MT) rs[2 * k + 1] = (float)Xh[k].y;
            }
            wave_sum_rows<MT>(rs);                                 // X.sum(dim=1), fp32 (dead columns hold zeros)
            red.row_steps(rs, fm);                                 // fold the waves, (sum - 1) / m once per row
            bool moved = false;
#pragma unroll
            for (int k = 0; k < MP; ++k) {
                const float t0 = (live && DMM_ROWH(2 * k)) ? rs[2 * k] : 0.0f;
                const float t1 = (live && 2 * k + 1 < MT && DMM_ROWH(2 * k + 1)) ? rs[2 * k + 1 < MT ? 2 * k + 1 : 0] : 0.0f;
                const h16x2 trp = {(_Float16)t0, (_Float16)t1};
                const h16x2 x = Xh[k];
                const h16x2 y = x - trp;                           // project_row (:9-19)
                P2[k] = x - y;
                Xh[k] = y;
                const h16x2 d = y - Xs[k];
                moved |= (d.x != (_Float16)0.0f) | (d.y != (_Float16)0.0f);
            }
            float mv[1] = {__ballot(moved) != 0ull ? 1.0f : 0.0f};
            red.fold(mv, fmax_op());
            if (mv[0] == 0.0f) break;                              // (:88-89)
        }
        if (cost_prev == cost) break;                              // (:96-98)
        cost_prev = cost;
    }
#undef DMM_ROWH
#pragma unroll
    for (int k = 0; k < MP; ++k) {
        X[2 * k] = (float)Xh[k].x;
        if (2 * k + 1 < MT) X[2 * k + 1] = (float)Xh[k].y;
    }
    return len - 1;                                    // sum(X_list) of column `col` stays in accbuf[i * LD + threadIdx.x]
}

// ---------------------------------------------------------------------------------------------
// Layer kernel: iou + mix with the cosine table + pad + solver + scores.  grid = B, block = 64*NG.
// ---------------------------------------------------------------------------------------------
// One frame: red_buf [2 * NG * (MT + 1)], xbuf [MT * 64 * NG], rsbuf [MT + 1] floats of LDS.
// bit 1 of the kernels' `is_test` argument: clear the count tables after reading them (internal; set only by
// relax_match_launch for the kernels built on relax_match_body)
constexpr int kRelaxClearTables = 2;

template <int MT, int NG, bool EXACT, bool HALF = false, bool W1TAPE = false>
__device__ __forceinline__ void relax_match_body(
    const float *__restrict__ cos_in, const int32_t *__restrict__ inter, const int32_t *__restrict__ area_p,
    const int32_t *__restrict__ area_t, const float *__restrict__ score_p, int N, int M,
    const int32_t *__restrict__ n_valid, const int32_t *__restrict__ m_valid, float w_feat, float w_iou,
    RelaxParams prm, int is_test, float *__restrict__ sim_out, float *__restrict__ R_out, float *__restrict__ Rb_out,
    float *__restrict__ match_score, float *__restrict__ det_score, int32_t *__restrict__ iters_out,
    float *__restrict__ X_final, float *red_buf, float *xbuf, float *rsbuf, int *hs,
    uint2 *__restrict__ tape_bits = nullptr, int *__restrict__ tape_sweeps = nullptr) {
    // tape_bits / tape_sweeps (W1TAPE kernels only; dmm_match_train_forward): [B][max_iter * proj_iter][64] sweep records and
    // [B][max_iter] executed-sweep counts for dmm_relax_match_bwd's taped form
    const int b = blockIdx.x;
    const int col = threadIdx.x;
    BlockRed<MT, NG> red(red_buf, threadIdx.x >> 6);
    const int Nb = n_valid ? n_valid[b] : N;
    const int Mb = EXACT ? MT : (m_valid ? m_valid[b] : M);
    const int PpS = N > M ? N : M + 1;                          // table stride
    float *Rb_b = Rb_out + (int64_t)b * M * PpS;
    float *R_b = R_out ? R_out + (int64_t)b * M * PpS : nullptr;
    float *X_b = X_final ? X_final + (int64_t)b * M * PpS : nullptr;
    float *sim_b = sim_out + (int64_t)b * M * N;
    if (Mb <= 0 || Nb <= 0) {                                   // dead frame: zeros (dmm_model.py:118-122)
        for (int i = threadIdx.x; i < M * PpS; i += 64 * NG) {
            Rb_b[i] = 0.0f;
            if (R_b) R_b[i] = 0.0f;
            if (X_b) X_b[i] = 0.0f;
        }
        for (int i = threadIdx.x; i < M * N; i += 64 * NG) sim_b[i] = 0.0f;
        for (int i = threadIdx.x; i < M; i += 64 * NG) {
            match_score[(int64_t)b * M + i] = 0.0f;
            det_score[(int64_t)b * M + i] = 0.0f;
        }
        if (iters_out && threadIdx.x == 0) iters_out[b] = 0;
        return;
    }
#define DMM_ROW(i) (EXACT || (i) < Mb)
    const int Pp = Nb > Mb ? Nb : Mb + 1;                       // live solver width (match_model.py:109-113)
    const bool has_prop = col < Nb;

    // ---- sim = (1-w)*cos + w*iou; pad; C = -sim ----
    float C[MT];
    {
        const float *cos_b = cos_in + (int64_t)b * M * N;
        const int32_t *inter_b = inter + (int64_t)b * M * N;
        const int ap = has_prop ? area_p[(int64_t)b * N + col] : 0;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float simv = 0.0f;
            C[i] = 0.0f;
            if (DMM_ROW(i) && has_prop) {
                const int in = inter_b[(int64_t)i * N + col];
                const int un = ap + area_t[(int64_t)b * M + i] - in;
                const float iou = (float)in / ((float)un + 1e-6f);     // match_helper.py:24-27
                const float a = cos_b[(int64_t)i * N + col] * w_feat, c = iou * w_iou;
                simv = a + c;                                          // match_model.py:90
                sim_b[(int64_t)i * N + col] = simv;
            }
            if (DMM_ROW(i) && col < Pp) C[i] = -simv;                  // padded columns: -0.0
        }
        // kRelaxClearTables (dmm_match_forward_ws, dense frames only): every table entry has been read by now -- each
        // inter / area_p entry by exactly one thread, area_t by every thread -- so the frame's tables go back to zero
        // here and the next call on this workspace starts its counts without a clearing launch in front.
        if (is_test & kRelaxClearTables) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the loads above have returned (same addresses below)
            if (NG > 1) __syncthreads();
            int32_t *wi = const_cast<int32_t *>(inter_b);
            if (has_prop) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    if (DMM_ROW(i)) wi[(int64_t)i * N + col] = 0;
                const_cast<int32_t *>(area_p)[(int64_t)b * N + col] = 0;
            }
            if (col < Mb) const_cast<int32_t *>(area_t)[(int64_t)b * M + col] = 0;
        }
        is_test &= 1;
    }

    float X[MT], acc[MT];
    int iters;
    const bool livec = col < Pp;
    if constexpr (HALF) {
        // 128-VGPR budget (4 waves per SIMD): nothing fp32-wide stays alive around the solver.  The sum of iterates comes
        // back in LDS (xbuf, column-major per thread), the final iterate is stored right here, and the epilogue reads the
        // cost back from sim (stored above; -C = sim_pad) one row at a time.
        iters = relax_core_h<MT, NG>(C, Mb, Pp, col, prm, red, xbuf, X);
        if (X_b) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
                if (DMM_ROW(i) && col < PpS) X_b[(int64_t)i * PpS + col] = livec ? X[i] : 0.0f;
        }
    } else {
        if constexpr (W1TAPE && NG == 1) {
            const RelaxTape tape{tape_bits + (size_t)b * prm.max_iter * prm.proj_iter * 64,
                                 tape_sweeps + (size_t)b * prm.max_iter};
            iters = relax_core<MT, NG, EXACT, false, true>(C, Mb, Pp, col, prm, red, xbuf, rsbuf, X, acc, nullptr, tape, hs);
        } else {
            iters = relax_core<MT, NG, EXACT>(C, Mb, Pp, col, prm, red, xbuf, rsbuf, X, acc, nullptr,
                                              RelaxTape{nullptr, nullptr}, hs);
        }
    }
    if (iters_out && threadIdx.x == 0) iters_out[b] = iters;

    // ---- R = sum(X_list)/len; logic; Rb; scores ----
    const float flen = (float)(iters + 1);
    const float sc = has_prop ? score_p[(int64_t)b * N + col] : 0.0f;
    float r[MT], rmax[MT], ms[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const float a = HALF ? xbuf[i * (64 * NG) + threadIdx.x] : acc[i];
        r[i] = a / flen;                                               // match_model.py:121
        rmax[i] = (livec && DMM_ROW(i)) ? r[i] : -__builtin_inff();
    }
    wave_max_rows<MT>(rmax);
    red.fold(rmax, fmax_op());                                         // (its barrier: every thread has read its sums)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const float lg = is_test ? (r[i] == rmax[i] ? 1.0f : 0.0f) : (r[i] > 0.01f ? 1.0f : 0.0f);
        const float rb = (livec && DMM_ROW(i)) ? r[i] * lg : 0.0f;     // :130
        const float rc = r[i] < 0.0f ? 0.0f : (r[i] > 1.0f ? 1.0f : r[i]);
        float simv;                                                    // = -C[i]: sim, +0 in the padded columns
        if constexpr (HALF) simv = (DMM_ROW(i) && has_prop) ? sim_b[(int64_t)i * N + col] : 0.0f;
        else simv = -C[i];
        ms[i] = (livec && DMM_ROW(i)) ? rc * simv : -__builtin_inff();       // :146
        const float ds = sc * rb;                                      // :147
        if (DMM_ROW(i) && livec) xbuf[i * Pp + col] = ds;
        if (DMM_ROW(i) && col < PpS) {
            Rb_b[(int64_t)i * PpS + col] = rb;
            if (R_b) R_b[(int64_t)i * PpS + col] = livec ? r[i] : 0.0f;
            if constexpr (!HALF) {
                if (X_b) X_b[(int64_t)i * PpS + col] = livec ? X[i] : 0.0f;
            }
        }
    }
    wave_max_rows<MT>(ms);
    red.fold(ms, fmax_op());
    row_sums_torch_order<NG>(xbuf, Mb, Pp, rsbuf);                     // (score * Rb).sum(1), ATen inner-sum order
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
            if (DMM_ROW(i)) {
                match_score[(int64_t)b * M + i] = ms[i];
                det_score[(int64_t)b * M + i] = rsbuf[i];
            }
    }
#undef DMM_ROW
    // rows of dead templates: zeros
    for (int i = Mb; i < M; ++i) {
        if (col < PpS) {
            Rb_b[(int64_t)i * PpS + col] = 0.0f;
            if (R_b) R_b[(int64_t)i * PpS + col] = 0.0f;
            if (X_b) X_b[(int64_t)i * PpS + col] = 0.0f;
        }
        if (col < N) sim_b[(int64_t)i * N + col] = 0.0f;
        if (threadIdx.x == 0) {
            match_score[(int64_t)b * M + i] = 0.0f;
            det_score[(int64_t)b * M + i] = 0.0f;
        }
    }
    // live rows, dead proposal columns of sim: zeros
    if (!has_prop && col < N)
        for (int i = 0; i < Mb; ++i) sim_b[(int64_t)i * N + col] = 0.0f;
}
}  // namespace dmm
