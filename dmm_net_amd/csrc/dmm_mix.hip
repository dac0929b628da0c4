// dmm_mix.hip -- assignment-weighted mask mix on gfx950.
//
// Replaces torch.mm(binary_Ridx_matched, pad_proposed_mask2d) of the reference
// (dmm/modules/match_model.py:134-144): full_outmask[m, :] = sum_n Rb[m, n] * mask_p[n, :].
// Rb is sparse by construction (test mode: the row maxima only; train mode: entries > 0.01), so
// only proposal planes with a non-zero weight are streamed.
//
// Roofline: HBM.  Bytes per frame = (#selected planes + M) * HW * 4  (test mode: <= 2*M*HW*4).
#include <stdlib.h>

#include <type_traits>

#include "dmm_common.h"

namespace dmm {

constexpr int kMixThreads = 256;
static thread_local bool g_drb_prezeroed = false;     // set (per thread, for one call) by mask_mix_bwd_prezeroed: dRb is zero already
static thread_local bool g_mix_shared_call = false;   // set by dmm_mask_mix_shared_* around the common entry point

// ---------------------------------------------------------------------------------------------
// One workgroup = one output row m of one frame over a pixel range.
// In test mode a row has exactly one weighted plane, so the kernel degenerates to a scaled copy with
// ONE read stream and ONE write stream per workgroup (a first version that streamed the union of the
// selected planes once and fanned it into all M rows kept ~20 DRAM streams per workgroup open and ran
// ~15 % slower).  Loads are issued kRowLoads at a time: G = 8 / cnt
// consecutive 4 KiB steps of the row's cnt planes.  Planes shared by several rows (train mode) are
// re-read per row; they are adjacent in time and mostly hit L2.
// ---------------------------------------------------------------------------------------------
constexpr int kRowLoads = 8;

// Output element types: fp32 (the nn.Module boundary) or the 16-bit storage type of the input planes (config 5: the
// matched masks become the next frame's fp16 templates; one rounding of the fp32 result).
template <typename TO> struct MixOut;
template <> struct MixOut<float> {
    template <int E, bool NT>
    static __device__ __forceinline__ void store(float *o, const float (&a)[E]) {
#pragma unroll
        for (int q = 0; q < E / 4; ++q) {
            float4a t;
            t.x = a[4 * q]; t.y = a[4 * q + 1]; t.z = a[4 * q + 2]; t.w = a[4 * q + 3];
            if (NT) __builtin_nontemporal_store(t, reinterpret_cast<float4a *>(o) + q);
            else reinterpret_cast<float4a *>(o)[q] = t;
        }
    }
    static __device__ __forceinline__ void store1(float *o, float v) { *o = v; }
};
template <> struct MixOut<f16_t> {
    typedef _Float16 half8a __attribute__((ext_vector_type(8)));
    template <int E, bool NT>
    static __device__ __forceinline__ void store(f16_t *o, const float (&a)[E]) {
        static_assert(E == 8, "16-bit outputs are written 8 pixels (16 bytes) per lane");
        half8a t;
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = (_Float16)a[k];
        if (NT) __builtin_nontemporal_store(t, reinterpret_cast<half8a *>(o));
        else *reinterpret_cast<half8a *>(o) = t;
    }
    static __device__ __forceinline__ void store1(f16_t *o, float v) { o->v = (_Float16)v; }
};
template <> struct MixOut<bf16_t> {
    typedef uint32_t uint4a __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ uint32_t rne(float v) {            // fp32 -> bfloat16 bits, round to nearest even
        const uint32_t u = __float_as_uint(v);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;   // NaN stays NaN
        return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    }
    template <int E, bool NT>
    static __device__ __forceinline__ void store(bf16_t *o, const float (&a)[E]) {
        static_assert(E == 8, "16-bit outputs are written 8 pixels (16 bytes) per lane");
        uint4a t;
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = rne(a[2 * k]) | (rne(a[2 * k + 1]) << 16);
        if (NT) __builtin_nontemporal_store(t, reinterpret_cast<uint4a *>(o));
        else *reinterpret_cast<uint4a *>(o) = t;
    }
    static __device__ __forceinline__ void store1(bf16_t *o, float v) { o->v = (uint16_t)rne(v); }
};

// E pixels of one plane at x (may stick out of [0, HW) at either end: those go element-wise, zeros outside)
template <typename T, int E, bool NT>
__device__ __forceinline__ void mix_load(const T *plane, int x, int HW, bool on, float (&v)[E]) {
    if (on && x >= 0 && x + E - 1 < HW) {
        if (E == 4) {
            float t[4];
            MaskIO<T>::template load4<NT>(plane + x, t);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = t[k];
        } else {
            typename MaskIO<T>::Raw r = MaskIO<T>::load_raw(plane + x);     // one 16-byte (8 x 16-bit) lane load
#pragma unroll
            for (int k = 0; k < E; ++k) v[k] = MaskIO<T>::elem(r, k);
        }
    } else {
#pragma unroll
        for (int k = 0; k < E; ++k) v[k] = (on && x + k >= 0 && x + k < HW) ? MaskIO<T>::load1(plane + x + k) : 0.0f;
    }
}

template <typename T, typename TO, int CNT, int NT>   // CNT = entries of this row if 1 or 2, 0 = generic; NT bit0 = nt loads, bit1 = nt stores
__device__ __forceinline__ void mix_row_range(const T *Pb, int64_t sp_n, const int *col_s, const float *w_s, int cnt,
                                              TO *orow, int HW, int pre, int s_begin, int s_end) {
    // A lane takes E = 16 bytes / sizeof(T) consecutive pixels (4 fp32 or 8 16-bit: always one full-width lane load).
    // Pixel index of thread t at step s is x = (256 s + t) * E - pre with pre = elements the row start lies past a
    // 128-byte line: every wave STORE then covers whole lines (rows of 65025 elements start on element boundaries
    // only; measured 5.78 TB/s with 16-byte-aligned stores, 6.03 with line-aligned ones), the loads take the
    // misalignment instead (free on gfx950, tools/hbm_probe.py).  The vectors that stick out of
    // [0, HW) at either end go element-wise.
    constexpr int E = 16 / (int)sizeof(T);
    constexpr int G = CNT == 0 ? 1 : 2;     // steps per iteration = the launcher's step quantum
    for (int s0 = s_begin; s0 < s_end; s0 += G) {
        float acc[G][E];
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int k = 0; k < E; ++k) acc[g][k] = 0.0f;
        if (CNT > 0) {
            float v[G][CNT][E];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int x = ((s0 + g) * kMixThreads + threadIdx.x) * E - pre;
#pragma unroll
                for (int e = 0; e < CNT; ++e)
                    mix_load<T, E, (NT & 1) != 0>(Pb + (int64_t)col_s[e] * sp_n, x, HW, s0 + g < s_end, v[g][e]);
            }
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int e = 0; e < CNT; ++e) {
                    const float w = w_s[e];
#pragma unroll
                    for (int k = 0; k < E; ++k) acc[g][k] = __builtin_fmaf(w, v[g][e][k], acc[g][k]);
                }
        } else {
            constexpr int UL = E == 4 ? kRowLoads : kRowLoads / 2;      // same bytes in flight per lane
            const int x = (s0 * kMixThreads + threadIdx.x) * E - pre;
            for (int e0 = 0; e0 < cnt; e0 += UL) {
                float v[UL][E];
#pragma unroll
                for (int u = 0; u < UL; ++u) {
                    const int e = e0 + u < cnt ? e0 + u : cnt - 1;
                    mix_load<T, E, (NT & 1) != 0>(Pb + (int64_t)col_s[e] * sp_n, x, HW, true, v[u]);
                }
#pragma unroll
                for (int u = 0; u < UL; ++u)
                    if (e0 + u < cnt) {
                        const float w = w_s[e0 + u];
#pragma unroll
                        for (int k = 0; k < E; ++k) acc[0][k] = __builtin_fmaf(w, v[u][k], acc[0][k]);
                    }
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int x = ((s0 + g) * kMixThreads + threadIdx.x) * E - pre;
            if (s0 + g >= s_end || x >= HW) continue;
            TO *o = orow + x;
            if (x >= 0 && x + E - 1 < HW) {
                MixOut<TO>::template store<E, (NT & 2) != 0>(o, acc[g]);
            } else {
#pragma unroll
                for (int k = 0; k < E; ++k)
                    if (x + k >= 0 && x + k < HW) MixOut<TO>::store1(o + k, acc[g][k]);
            }
        }
    }
}

// grid = (pixel splits, M, B)
template <typename T, typename TO, int NT>
__global__ __launch_bounds__(kMixThreads) void mask_mix_rows_kernel(const float *__restrict__ Rb,
                                                                    const T *__restrict__ masks_p, int N, int M, int Pp,
                                                                    int HW, int64_t sp_b, int64_t sp_n,
                                                                    const int32_t *__restrict__ n_valid,
                                                                    const int32_t *__restrict__ m_valid,
                                                                    TO *__restrict__ out, int64_t so_b, int64_t so_m,
                                                                    int steps_per_wg, int align_mask, int xcd_remap) {
    constexpr int E = 16 / (int)sizeof(T);
    __shared__ float w_s[DMM_MAX_PROPOSALS];
    __shared__ int col_s[DMM_MAX_PROPOSALS];
    __shared__ int wcnt_s[kMixThreads / 64];
    // XCD-aware mapping (workgroups go round-robin over the 8 XCDs by linear id): every complete group of 8 output rows
    // gives each XCD one whole row, so the misaligned source lines neighbouring pixel ranges share stay in one L2
    int b = blockIdx.z, m = blockIdx.y, range = blockIdx.x;
    if (xcd_remap) {
        const int splits = gridDim.x, rows = gridDim.y * gridDim.z;
        const int row = blockIdx.y + gridDim.y * blockIdx.z;
        if (row < (rows & ~7)) {
            const int id = blockIdx.x + splits * row;
            const int grp = id / (8 * splits), within = id - grp * 8 * splits;
            const int r2 = grp * 8 + (within & 7);
            range = within >> 3;
            m = r2 % (int)gridDim.y;
            b = r2 / (int)gridDim.y;
        }
    }
    int Nb = n_valid ? n_valid[b] : N;
    int Mb = m_valid ? m_valid[b] : M;
    if (Nb <= 0) Mb = 0;
    // The row's non-zero weights, compacted in column order.  ONE global round trip: thread n reads Rb[m, n] (N <=
    // DMM_MAX_PROPOSALS = the workgroup size), a ballot per wave and the wave counts through LDS give the positions.
    // (A 64-lane loop over the row was 4 dependent round trips at N = 200 in front of 16 KB of streaming per
    // workgroup: config 5's mix ran at 0.63 of peak against 0.78 for the N = 50 rows of config 2.)
    static_assert(kMixThreads >= DMM_MAX_PROPOSALS, "one thread per proposal column");
    {
        const int n = threadIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const float w = (m < Mb && n < Nb) ? Rb[((int64_t)b * M + m) * Pp + n] : 0.0f;
        const unsigned long long bal = __ballot(w != 0.0f);
        if (lane == 0) wcnt_s[wave] = __builtin_popcountll(bal);
        __syncthreads();
        int base = 0;
#pragma unroll
        for (int k = 0; k < kMixThreads / 64; ++k) base += k < wave ? wcnt_s[k] : 0;
        if (w != 0.0f) {
            const int pos = base + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
            col_s[pos] = n;
            w_s[pos] = w;
        }
    }
    __syncthreads();
    const int cnt = wcnt_s[0] + wcnt_s[1] + wcnt_s[2] + wcnt_s[3];
    const T *Pb = frame_base(masks_p, b, sp_b);
    TO *orow = out + (int64_t)b * so_b + (int64_t)m * so_m;
    // row start = 128-byte boundary + pre elements (align_mask = elements per aligned unit - 1)
    const int pre = (int)((reinterpret_cast<uintptr_t>(orow) / sizeof(TO)) & align_mask);
    const int nsteps = (HW + pre + kMixThreads * E - 1) / (kMixThreads * E);
    const int s_begin = range * steps_per_wg;
    const int s_end = min(nsteps, s_begin + steps_per_wg);
    if (cnt == 1) mix_row_range<T, TO, 1, NT>(Pb, sp_n, col_s, w_s, cnt, orow, HW, pre, s_begin, s_end);
    else if (cnt == 2) mix_row_range<T, TO, 2, NT>(Pb, sp_n, col_s, w_s, cnt, orow, HW, pre, s_begin, s_end);
    else mix_row_range<T, TO, 0, NT>(Pb, sp_n, col_s, w_s, cnt, orow, HW, pre, s_begin, s_end);   // cnt == 0 writes zeros
}

// ---------------------------------------------------------------------------------------------
// Backward of the mix with respect to Rb:  dRb[b,m,n] = sum_x dOut[b,m,x] * masks_p[b,n,x], needed only where
// the forward weight was non-zero (Rb = R * logic, match_model.py:130: logic is a constant 0/1 mask, so the
// gradient of every masked-out entry is dropped anyway).  Same row-major decomposition as the forward:
// one workgroup = one template row over a pixel range, it streams dOut[m] once per batch of 8 selected planes
// and finishes with one fp32 atomic per (entry, workgroup) into dRb (zeroed by the launcher).
// The reference gets this from torch.mm's autograd (a dense [O,HW] x [HW,P] product reading all P planes).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kMixThreads) void mask_mix_bwd_kernel(const float *__restrict__ Rb,
                                                                   const T *__restrict__ masks_p,
                                                                   const float *__restrict__ dout, int N, int M, int Pp,
                                                                   int HW, int64_t sp_b, int64_t sp_n,
                                                                   const int32_t *__restrict__ n_valid,
                                                                   const int32_t *__restrict__ m_valid,
                                                                   float *__restrict__ dRb, int steps_per_wg) {
    __shared__ int col_s[DMM_MAX_PROPOSALS];
    __shared__ float part_s[4][kRowLoads];
    __shared__ int cnt_s;
    const int b = blockIdx.z, m = blockIdx.y;
    int Nb = n_valid ? n_valid[b] : N;
    int Mb = m_valid ? m_valid[b] : M;
    if (Nb <= 0) Mb = 0;
    if (m >= Mb) return;
    if (threadIdx.x < 64) {
        int base = 0;
        const float *Rrow = Rb + ((int64_t)b * M + m) * Pp;
        for (int n0 = 0; n0 < Nb; n0 += 64) {
            const int n = n0 + threadIdx.x;
            const bool nz = n < Nb && Rrow[n] != 0.0f;
            const unsigned long long bal = __ballot(nz);
            if (nz) col_s[base + __builtin_popcountll(bal & ((1ull << threadIdx.x) - 1ull))] = n;
            base += __builtin_popcountll(bal);
        }
        if (threadIdx.x == 0) cnt_s = base;
    }
    __syncthreads();
    const int cnt = cnt_s;
    if (cnt == 0) return;
    const T *Pb = frame_base(masks_p, b, sp_b);
    const float *drow = dout + ((int64_t)b * M + m) * HW;
    const int nsteps = (HW + kMixThreads * 4 - 1) / (kMixThreads * 4);
    const int s_begin = blockIdx.x * steps_per_wg;
    const int s_end = min(nsteps, s_begin + steps_per_wg);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int e0 = 0; e0 < cnt; e0 += kRowLoads) {
        float acc[kRowLoads];
#pragma unroll
        for (int u = 0; u < kRowLoads; ++u) acc[u] = 0.0f;
        for (int s = s_begin; s < s_end; ++s) {
            const int x = (s * kMixThreads + threadIdx.x) * 4;
            float d[4];
            if (x + 3 < HW) {
                const float4u t = *reinterpret_cast<const float4u *>(drow + x);
                d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) d[k] = x + k < HW ? drow[x + k] : 0.0f;
            }
            float v[kRowLoads][4];
#pragma unroll
            for (int u = 0; u < kRowLoads; ++u) {
                const int e = e0 + u < cnt ? e0 + u : cnt - 1;
                const T *plane = Pb + (int64_t)col_s[e] * sp_n;
                if (x + 3 < HW) {
                    MaskIO<T>::load4(plane + x, v[u]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[u][k] = x + k < HW ? MaskIO<T>::load1(plane + x + k) : 0.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < kRowLoads; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[u] = __builtin_fmaf(d[k], v[u][k], acc[u]);
        }
        wave_sum_rows<kRowLoads>(acc);
        __syncthreads();
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < kRowLoads; ++u) part_s[wave][u] = acc[u];
        }
        __syncthreads();
        if (threadIdx.x < kRowLoads && e0 + threadIdx.x < cnt) {
            const float t = ((part_s[0][threadIdx.x] + part_s[1][threadIdx.x]) + part_s[2][threadIdx.x]) + part_s[3][threadIdx.x];
            atomicAdd(&dRb[((int64_t)b * M + m) * Pp + col_s[e0 + threadIdx.x]], t);
        }
    }
}

template <typename T>
static int mask_mix_bwd_typed(const float *Rb, const T *masks_p, const float *dout, int B, int N, int M, int Pp, int HW,
                              int64_t sp_b, int64_t sp_n, const int32_t *n_valid, const int32_t *m_valid, float *dRb,
                              hipStream_t stream) {
    if (!g_drb_prezeroed) DMM_HIP_TRY(zero_async(dRb, sizeof(float) * (size_t)B * M * Pp, stream));
    const int nsteps = (HW + kMixThreads * 4 - 1) / (kMixThreads * 4);
    int splits = (8192 + B * M - 1) / (B * M);
    if (splits > nsteps) splits = nsteps;
    if (splits < 1) splits = 1;
    const int steps_per_wg = (nsteps + splits - 1) / splits;
    splits = (nsteps + steps_per_wg - 1) / steps_per_wg;
    hipLaunchKernelGGL((mask_mix_bwd_kernel<T>), dim3(splits, M, B), dim3(kMixThreads), 0, stream, Rb, masks_p, dout, N, M,
                       Pp, HW, sp_b, sp_n, n_valid, m_valid, dRb, steps_per_wg);
    return check_launch();
}

template <typename T, typename TO>
static int mask_mix_typed(const float *Rb, const T *masks_p, int B, int N, int M, int Pp, int HW, int64_t sp_b,
                          int64_t sp_n, const int32_t *n_valid, const int32_t *m_valid, TO *out, int64_t so_b,
                          int64_t so_m, hipStream_t stream) {
    constexpr int E = 16 / (int)sizeof(T);
    const int align_bytes = opt(DMM_OPT_MIX_ALIGN);
    const int align_mask = align_bytes / (int)sizeof(TO) - 1;
    const int nsteps = (HW + align_mask + kMixThreads * E - 1) / (kMixThreads * E);    // worst-case row misalignment
    // MANY TINY workgroups: 2 steps = 8 KiB of the row each, up to ~320k of them.  Measured at B = 1024 (test mode,
    // one plane per row): 4.2 / 4.9 / 5.1 / 5.2 / 5.75-6.1 TB/s at 10k / 40k / 80k / 160k / 320k workgroups; 1-step
    // workgroups fall back to 5.5-5.8.  In dispatch order the resident workgroups then cover a compact, advancing
    // window of the output instead of ~2000 independent 64 KiB streams.  DMM_OPT_MIX_WGS / DMM_OPT_MIX_STEPQ override.
    const int target_wgs = opt(DMM_OPT_MIX_WGS);
    int splits = (target_wgs + B * M - 1) / (B * M);
    const int step_q = opt(DMM_OPT_MIX_STEPQ);
    const int max_splits = (nsteps + step_q - 1) / step_q;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    const int steps_per_wg = ((nsteps + splits - 1) / splits + step_q - 1) / step_q * step_q;
    splits = (nsteps + steps_per_wg - 1) / steps_per_wg;
    const int nt_mode = opt(DMM_OPT_MIX_NT);
    const int xcd_remap = opt(DMM_OPT_MIX_XCD);
#define DMM_MIX_LAUNCH(NT)                                                                                              \
    hipLaunchKernelGGL((mask_mix_rows_kernel<T, TO, NT>), dim3(splits, M, B), dim3(kMixThreads), 0, stream, Rb, masks_p, \
                       N, M, Pp, HW, sp_b, sp_n, n_valid, m_valid, out, so_b, so_m, steps_per_wg, align_mask, xcd_remap)
    switch (nt_mode & 3) {
        case 0: DMM_MIX_LAUNCH(0); break;
        case 1: DMM_MIX_LAUNCH(1); break;
        case 2: DMM_MIX_LAUNCH(2); break;
        default: DMM_MIX_LAUNCH(3); break;
    }
#undef DMM_MIX_LAUNCH
    return check_launch();
}

// ---------------------------------------------------------------------------------------------
// Rows that SHARE planes (train mode: logic = R > 0.01, match_model.py:126-129 -- at BASELINE configs[1] every template
// row keeps ~13 of the 50 proposals, 133 (row, plane) pairs per frame over 50 distinct planes).  The row kernel above
// streams a plane once per ROW that uses it; here one workgroup owns a pixel range of a frame, streams every plane of
// the UNION of the rows' supports exactly once and fans it into the rows' accumulators: (|union| + M) planes of traffic
// instead of (pairs + M).  Per row the arithmetic is the row kernel's -- acc = fma(w, v, acc) from zero over the row's
// non-zero weights in ascending column order -- so both kernels agree bit for bit.  Test mode (one plane per row, nothing
// shared) stays on the row kernel: one read and one write stream per workgroup beat 2 M streams there (-15 %).
// grid = (pixel splits, B); LDS: the union's columns, a row bit mask per column, the weights [column][row].
// ---------------------------------------------------------------------------------------------
constexpr int kSharedLoads = 8;                                           // planes in flight per lane

// the union of the rows' supports, compacted in column order (one thread per proposal column, one global round trip per
// row); returns the number of union columns.  rowmask_s[e] bit m = row m uses column col_s[e].
template <int MT>
__device__ __forceinline__ int shared_support(const float *__restrict__ Rb_b, int Pp, int Nb, int Mb, int *col_s,
                                              unsigned *rowmask_s, float *w_c, int *wcnt_s) {
    const int n = threadIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float w[MT];
    unsigned mask = 0u;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        w[m] = (m < Mb && n < Nb) ? Rb_b[(int64_t)m * Pp + n] : 0.0f;
        mask |= (w[m] != 0.0f) ? (1u << m) : 0u;
    }
    const unsigned long long bal = __ballot(mask != 0u);
    if (lane == 0) wcnt_s[wave] = __builtin_popcountll(bal);
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int k = 0; k < kMixThreads / 64; ++k) base += k < wave ? wcnt_s[k] : 0;
    if (mask != 0u) {
        const int pos = base + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
        col_s[pos] = n;
        rowmask_s[pos] = mask;
        if (w_c) {
#pragma unroll
            for (int m = 0; m < MT; ++m) w_c[pos * MT + m] = w[m];
        }
    }
    __syncthreads();
    return wcnt_s[0] + wcnt_s[1] + wcnt_s[2] + wcnt_s[3];
}

__device__ __forceinline__ unsigned uniform_u32(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ float uniform_f32(float v) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

template <typename T, typename TO, int MT, int NT>
__global__ __launch_bounds__(kMixThreads) void mask_mix_shared_kernel(const float *__restrict__ Rb,
                                                                      const T *__restrict__ masks_p, int N, int M, int Pp,
                                                                      int HW, int64_t sp_b, int64_t sp_n,
                                                                      const int32_t *__restrict__ n_valid,
                                                                      const int32_t *__restrict__ m_valid,
                                                                      TO *__restrict__ out, int64_t so_b, int64_t so_m,
                                                                      int steps_per_wg, int lockstep, int xcd_remap) {
    constexpr int E = 4;                                                  // pixels per lane and step (16 bytes of fp32)
    __shared__ __attribute__((aligned(16))) float w_c[DMM_MAX_PROPOSALS * MT];
    __shared__ int col_s[DMM_MAX_PROPOSALS];
    __shared__ __attribute__((aligned(16))) unsigned rowmask_s[DMM_MAX_PROPOSALS];
    __shared__ int wcnt_s[kMixThreads / 64];
    // DMM_OPT_MIX_XCD bit 1: every complete group of 8 frames gives each XCD one whole frame (dmm_common.h): the 128-byte
    // lines that neighbouring 4 KiB steps of a plane share are asked for by ONE L2, a few dispatches apart.  50 x 10 train-mode
    // supports, 512 frames: 1.364 -> 1.319 ms (0.706 -> 0.730 of 8 TB/s), traffic 1.041x -> 1.027x (tools/mix_bwd_probe.py).
    // The backward (bit 2, off) takes four steps per workgroup -- its neighbours are its own: 1.264 -> 1.270 ms, same traffic
    int b, range;
    xcd_frame_range(xcd_remap, b, range);
    int Nb = n_valid ? n_valid[b] : N;
    int Mb = m_valid ? m_valid[b] : M;
    if (Nb <= 0) Mb = 0;
    const int cnt = shared_support<MT>(Rb + (int64_t)b * M * Pp, Pp, Nb, Mb, col_s, rowmask_s, w_c, wcnt_s);
    const T *Pb = frame_base(masks_p, b, sp_b);
    TO *ob = out + (int64_t)b * so_b;
    const int nsteps = (HW + kMixThreads * E - 1) / (kMixThreads * E);
    const int s_begin = range * steps_per_wg;
    const int s_end = min(nsteps, s_begin + steps_per_wg);
    for (int s = s_begin; s < s_end; ++s) {
        const int x = (s * kMixThreads + threadIdx.x) * E;
        float acc[MT][E];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int k = 0; k < E; ++k) acc[m][k] = 0.0f;
        for (int e0 = 0; e0 < cnt; e0 += kSharedLoads) {
            // DMM_OPT_MIX_SHARED_LOCKSTEP: the four waves take the four neighbouring 1 KiB pieces of a plane; kept in step
            // (one barrier per group of planes) they ask for the 128-byte lines their pieces share at the same time
            if (lockstep) __syncthreads();
            float v[kSharedLoads][E];
#pragma unroll
            for (int u = 0; u < kSharedLoads; ++u) {
                const int e = e0 + u < cnt ? e0 + u : cnt - 1;
                mix_load<T, E, (NT & 1) != 0>(Pb + (int64_t)col_s[e] * sp_n, x, HW, true, v[u]);
            }
            // Weights and row masks come out of LDS one at a time through readfirstlane (scalar operands, scalar branches).
            // Tried and measured SLOWER at B = 512, 50 x 10 (1.57 ms): a column's weights as vector LDS reads with 4 planes
            // in flight (the accumulators + 8 planes + MT weights do not fit 128 VGPRs): 1.80 ms.  The kernel sits at the
            // ceiling of its ACCESS PATTERN, not of its instruction stream: tools/mix_shared_probe.py streams 48 planes +
            // 10 written rows per frame with no arithmetic at 5.1 TB/s (6.4 read-only).
#pragma unroll
            for (int u = 0; u < kSharedLoads; ++u) {
                // wave-uniform: scalar branches per row (0 = past the end of the union)
                const unsigned rows = e0 + u < cnt ? uniform_u32(rowmask_s[e0 + u]) : 0u;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    if (rows & (1u << m)) {
                        const float w = uniform_f32(w_c[(e0 + u) * MT + m]);
#pragma unroll
                        for (int k = 0; k < E; ++k) acc[m][k] = __builtin_fmaf(w, v[u][k], acc[m][k]);
                    }
                }
            }
        }
        if (x >= HW) continue;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (m >= M) continue;
            TO *o = ob + (int64_t)m * so_m + x;
            if (x + E - 1 < HW) {
                if (sizeof(TO) == 4) {
                    float4u t;
                    t.x = acc[m][0]; t.y = acc[m][1]; t.z = acc[m][2]; t.w = acc[m][3];
                    if (NT & 2) __builtin_nontemporal_store(t, reinterpret_cast<float4u *>(o));
                    else *reinterpret_cast<float4u *>(o) = t;
                } else {
#pragma unroll
                    for (int k = 0; k < E; ++k) MixOut<TO>::store1(o + k, acc[m][k]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < E; ++k)
                    if (x + k < HW) MixOut<TO>::store1(o + k, acc[m][k]);
            }
        }
    }
}

template <typename T, typename TO>
static int mask_mix_shared_typed(const float *Rb, const T *masks_p, int B, int N, int M, int Pp, int HW, int64_t sp_b,
                                 int64_t sp_n, const int32_t *n_valid, const int32_t *m_valid, TO *out, int64_t so_b,
                                 int64_t so_m, hipStream_t stream) {
    const int nsteps = (HW + kMixThreads * 4 - 1) / (kMixThreads * 4);
    // ONE step (4 KiB of every plane of the union) per workgroup: the access-pattern probe (tools/mix_shared_probe.py) loses
    // 4 % / 8 % with 2 / 4 steps per workgroup, whatever the batch (DMM_OPT_MIX_SHARED_STEPS)
    int steps_per_wg = opt(DMM_OPT_MIX_SHARED_STEPS);
    if (steps_per_wg < 1) steps_per_wg = 1;
    const int splits = (nsteps + steps_per_wg - 1) / steps_per_wg;
    const int nt_mode = opt(DMM_OPT_MIX_NT);
#define DMM_MIXS_LAUNCH(MT_, NT_)                                                                                       \
    hipLaunchKernelGGL((mask_mix_shared_kernel<T, TO, MT_, NT_>), dim3(splits, B), dim3(kMixThreads), 0, stream, Rb,    \
                       masks_p, N, M, Pp, HW, sp_b, sp_n, n_valid, m_valid, out, so_b, so_m, steps_per_wg,               \
                       opt(DMM_OPT_MIX_SHARED_LOCKSTEP), (opt(DMM_OPT_MIX_XCD) >> 1) & 1)
#define DMM_MIXS_PICK(MT_)                       \
    do {                                         \
        if ((nt_mode & 3) == 3) DMM_MIXS_LAUNCH(MT_, 3); \
        else DMM_MIXS_LAUNCH(MT_, 0);            \
    } while (0)
    if (M <= 8) DMM_MIXS_PICK(8);
    else if (M <= 16) DMM_MIXS_PICK(16);
    else DMM_MIXS_PICK(32);
#undef DMM_MIXS_PICK
#undef DMM_MIXS_LAUNCH
    return check_launch();
}

// Backward of the mix, rows sharing planes: one workgroup = a pixel range of a frame; d full_outmask of the M rows is
// loaded once per step (registers), every plane of the union once.  Each (row, plane) pair of the support has a SLOT
// (pairs in column-major order, <= kPairSlots per frame: more fall back to a dense table).  A pair's 4-pixel dot product
// is reduced inside each 16-lane row of the wave (four DPP adds) and the four row sums go into the wave's LDS slab with ONE
// ds_add_f32 (lanes 0 / 16 / 32 / 48, four neighbouring words: no same-address serialisation) -- round 5; the round-4 form
// finished the reduction across the rows (two more DPP steps behind broadcast moves), read the total back into a scalar
// and parked it in "lane slot of register slot / 64" by selecting over all four accumulator registers: 28 VALU instructions
// and ~10 hazard no-ops per pair against 9 here (profiles/r05_mix_bwd_*).  At the end the four waves' slabs are folded in
// a fixed order and one global atomic per pair and workgroup goes to dRb.  (pairs + M) -> (|union| + M) planes of traffic.
constexpr int kPairSlots = 256;

template <typename T, int MT>
__global__ __launch_bounds__(kMixThreads) void mask_mix_bwd_shared_kernel(const float *__restrict__ Rb,
                                                                          const T *__restrict__ masks_p,
                                                                          const float *__restrict__ dout, int N, int M,
                                                                          int Pp, int HW, int64_t sp_b, int64_t sp_n,
                                                                          const int32_t *__restrict__ n_valid,
                                                                          const int32_t *__restrict__ m_valid,
                                                                          float *__restrict__ dRb, int steps_per_wg,
                                                                          int lockstep, int xcd_remap) {
    constexpr int E = 4;
    __shared__ int col_s[DMM_MAX_PROPOSALS];
    __shared__ __attribute__((aligned(16))) unsigned rowmask_s[DMM_MAX_PROPOSALS];
    __shared__ int pbase_s[DMM_MAX_PROPOSALS + 1];                        // first slot of a union column
    // ONE dynamic block serves both forms (they are mutually exclusive: pairs <= kPairSlots or not): the slot slabs
    // [wave][slot][16-lane row] = 16 KB, or the dense per-wave tables 4 * N * MT floats.  The launcher sizes it as the
    // larger of the two, so a workgroup's LDS is that + ~3.1 KB of static tables: under 64 KB for everything the gate in
    // mask_mix_bwd admits (as two separate arrays the worst case was ~75 KB).
    extern __shared__ __attribute__((aligned(16))) float dyn_s[];
    float (*fold_s)[kPairSlots][4] = reinterpret_cast<float (*)[kPairSlots][4]>(dyn_s);
    __shared__ int wcnt_s[kMixThreads / 64];
    int b, range;
    xcd_frame_range(xcd_remap, b, range);                                 // DMM_OPT_MIX_XCD bit 2; see mask_mix_shared_kernel
    int Nb = n_valid ? n_valid[b] : N;
    int Mb = m_valid ? m_valid[b] : M;
    if (Nb <= 0) Mb = 0;
    if (Mb <= 0) return;
    const int cnt = shared_support<MT>(Rb + (int64_t)b * M * Pp, Pp, Nb, Mb, col_s, rowmask_s, (float *)nullptr, wcnt_s);
    if (cnt == 0) return;
    {                                                                     // exclusive prefix sum of the columns' pair counts
        const int e = threadIdx.x, wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
        const int pc = e < cnt ? __builtin_popcount(rowmask_s[e]) : 0;
        int inc = pc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(inc, d);
            if (ln >= d) inc += t;
        }
        __syncthreads();                                                  // wcnt_s is free again (shared_support read it)
        if (ln == 63) wcnt_s[wv] = inc;
        __syncthreads();
        int base = 0;
#pragma unroll
        for (int k = 0; k < kMixThreads / 64; ++k) base += k < wv ? wcnt_s[k] : 0;
        if (e < cnt) pbase_s[e] = base + inc - pc;
        if (e == cnt - 1) pbase_s[cnt] = base + inc;
    }
    __syncthreads();
    const int pairs = pbase_s[cnt];
    const T *Pb = frame_base(masks_p, b, sp_b);
    const float *db = dout + (int64_t)b * M * HW;
    const int nsteps = (HW + kMixThreads * E - 1) / (kMixThreads * E);
    const int s_begin = range * steps_per_wg;
    const int s_end = min(nsteps, s_begin + steps_per_wg);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    typedef unsigned uint4v __attribute__((ext_vector_type(4)));
    if (pairs > kPairSlots) {
        // a dense support (more pairs than slots): pair sums in LDS, one table PER WAVE ([wave][union column][row], dynamic
        // LDS: 4 * N * MT floats) -- lane 0 of a wave adds to its own table in program order, the tables are folded in a
        // fixed order.  One LDS round trip per pair: slower, and only here.
        float *acc_s = dyn_s;
        const int tbl = N * MT;
        for (int i = threadIdx.x; i < (kMixThreads / 64) * tbl; i += kMixThreads) acc_s[i] = 0.0f;
        __syncthreads();
        float *acc_w = acc_s + wave * tbl;
        for (int s = s_begin; s < s_end; ++s) {
            const int x = (s * kMixThreads + threadIdx.x) * E;
            for (int e = 0; e < cnt; ++e) {
                float v[E];
                mix_load<T, E, true>(Pb + (int64_t)col_s[e] * sp_n, x, HW, true, v);
                const unsigned rows = uniform_u32(rowmask_s[e]);
                for (int m = 0; m < Mb; ++m) {
                    if (!(rows & (1u << m))) continue;
                    float p = 0.0f;
#pragma unroll
                    for (int k = 0; k < E; ++k) {
                        const float dv = x + k < HW ? db[(int64_t)m * HW + x + k] : 0.0f;
                        p = k == 0 ? dv * v[0] : __builtin_fmaf(dv, v[k], p);
                    }
                    p = wave_sum(p);
                    if (lane == 0) acc_w[e * MT + m] += p;
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < cnt * MT; i += kMixThreads) {
            const int e = i / MT, m = i - e * MT;
            if (rowmask_s[e] & (1u << m))
                atomicAdd(&dRb[((int64_t)b * M + m) * Pp + col_s[e]],
                          ((acc_s[i] + acc_s[tbl + i]) + acc_s[2 * tbl + i]) + acc_s[3 * tbl + i]);
        }
        return;
    }
    for (int i = threadIdx.x; i < (kMixThreads / 64) * kPairSlots * 4; i += kMixThreads) (&fold_s[0][0][0])[i] = 0.0f;
    __syncthreads();
    float *acc_w = &fold_s[wave][0][0] + (lane >> 4);                     // + 4 * slot: this lane's row word of a slot
    const bool row_leader = (lane & 15) == 0;
    for (int s = s_begin; s < s_end; ++s) {
        const int x = (s * kMixThreads + threadIdx.x) * E;
        float d[MT][E];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (m < Mb && x + E - 1 < HW) {                  // (streamed once like the planes: non-temporal)
                const float4u t = __builtin_nontemporal_load(reinterpret_cast<const float4u *>(db + (int64_t)m * HW + x));
                d[m][0] = t.x; d[m][1] = t.y; d[m][2] = t.z; d[m][3] = t.w;
            } else {
#pragma unroll
                for (int k = 0; k < E; ++k) d[m][k] = (m < Mb && x + k < HW) ? db[(int64_t)m * HW + x + k] : 0.0f;
            }
        }
        for (int e0 = 0; e0 < cnt; e0 += kSharedLoads) {
            if (lockstep) __syncthreads();                                 // see mask_mix_shared_kernel
            float v[kSharedLoads][E];
#pragma unroll
            for (int u = 0; u < kSharedLoads; ++u) {
                const int e = e0 + u < cnt ? e0 + u : cnt - 1;
                mix_load<T, E, true>(Pb + (int64_t)col_s[e] * sp_n, x, HW, true, v[u]);
            }
            const uint4v ma = *reinterpret_cast<const uint4v *>(&rowmask_s[e0]);
            const uint4v mb = *reinterpret_cast<const uint4v *>(&rowmask_s[e0 + 4]);
            int slot = __builtin_amdgcn_readfirstlane(pbase_s[e0]);
#pragma unroll
            for (int u = 0; u < kSharedLoads; ++u) {
                const unsigned rows = e0 + u < cnt ? uniform_u32(u < 4 ? ma[u & 3] : mb[u & 3]) : 0u;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    if (rows & (1u << m)) {                                // wave-uniform
                        float p = d[m][0] * v[u][0];
                        p = __builtin_fmaf(d[m][1], v[u][1], p);
                        p = __builtin_fmaf(d[m][2], v[u][2], p);
                        p = __builtin_fmaf(d[m][3], v[u][3], p);
                        // the sum of each 16-lane row in all of its lanes (the first four steps of wave_sum)
                        p = p + dpp_full_f32<DPP_XOR1>(p);
                        p = p + dpp_full_f32<DPP_XOR2>(p);
                        p = p + dpp_full_f32<DPP_HALF_MIRROR>(p);
                        p = p + dpp_full_f32<DPP_MIRROR>(p);
                        if (row_leader) atomicAdd(acc_w + 4 * slot, p);   // ds_add_f32, no return: 4 lanes, 4 words
                        ++slot;
                    }
                }
            }
        }
    }
    __syncthreads();
    // slot -> (column, row): walk the union again, one thread per slot; waves and rows folded in a fixed order
    for (int i = threadIdx.x; i < pairs; i += kMixThreads) {
        int lo = 0, hi = cnt;                                             // pbase_s[lo] <= i < pbase_s[lo + 1]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pbase_s[mid] <= i) lo = mid; else hi = mid;
        }
        unsigned rows = rowmask_s[lo];
        for (int k = i - pbase_s[lo]; k > 0; --k) rows &= rows - 1;       // drop the k lowest set bits
        const int m = __builtin_ctz(rows);
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < kMixThreads / 64; ++w) {
            const float4 q = *reinterpret_cast<const float4 *>(&fold_s[w][i][0]);
            t = t + ((q.x + q.y) + (q.z + q.w));
        }
        atomicAdd(&dRb[((int64_t)b * M + m) * Pp + col_s[lo]], t);
    }
}

// dynamic LDS of mask_mix_bwd_shared_kernel: the slot slabs or the dense per-wave tables, whichever is larger
static inline size_t mix_bwd_dynamic_lds(int N, int mt) {
    const size_t slabs = sizeof(float) * (kMixThreads / 64) * kPairSlots * 4;
    const size_t dense = sizeof(float) * (kMixThreads / 64) * (size_t)N * mt;
    return dense > slabs ? dense : slabs;
}

template <typename T>
static int mask_mix_bwd_shared_typed(const float *Rb, const T *masks_p, const float *dout, int B, int N, int M, int Pp,
                                     int HW, int64_t sp_b, int64_t sp_n, const int32_t *n_valid, const int32_t *m_valid,
                                     float *dRb, hipStream_t stream) {
    if (!g_drb_prezeroed) DMM_HIP_TRY(zero_async(dRb, sizeof(float) * (size_t)B * M * Pp, stream));
    const int nsteps = (HW + kMixThreads * 4 - 1) / (kMixThreads * 4);
    // four steps per workgroup (the forward: one): every workgroup clears and folds its LDS slabs and ends with one atomic per
    // pair -- measured at B = 512, 50 x 10 with the waves in lock step: 1.281 / 1.268 ms at 2 / 4 steps (free running: 1.293 /
    // 1.329; the forward 1.387 / 1.430 at 1 / 2)
    int steps_per_wg = 4 * opt(DMM_OPT_MIX_SHARED_STEPS);
    // ... while that still leaves ~1024 workgroups: a one-frame call (112 steps at 255 x 448) took 64 us as 28 workgroups
    const int64_t fill = ((int64_t)B * nsteps + 1023) / 1024;
    if (steps_per_wg > fill) steps_per_wg = (int)fill;
    if (steps_per_wg < 1) steps_per_wg = 1;
    const int splits = (nsteps + steps_per_wg - 1) / steps_per_wg;
#define DMM_MIXB_LAUNCH(MT_)                                                                                        \
    hipLaunchKernelGGL((mask_mix_bwd_shared_kernel<T, MT_>), dim3(splits, B), dim3(kMixThreads),                    \
                       mix_bwd_dynamic_lds(N, MT_), stream, Rb, masks_p, dout, N, M, Pp, HW,                         \
                       sp_b, sp_n, n_valid, m_valid, dRb, steps_per_wg, opt(DMM_OPT_MIX_SHARED_LOCKSTEP),               \
                       (opt(DMM_OPT_MIX_XCD) >> 2) & 1)
    if (M <= 8) DMM_MIXB_LAUNCH(8);
    else if (M <= 16) DMM_MIXB_LAUNCH(16);
    else DMM_MIXB_LAUNCH(32);
#undef DMM_MIXB_LAUNCH
    return check_launch();
}

// ---------------------------------------------------------------------------------------------
// The mix for ANY N, M (tables outside the fast kernel's envelope: it compacts a row's weights into a 256-entry LDS list
// with one thread per proposal column).  Same arithmetic -- the non-zero weights of the row in ascending column order,
// acc = fma(w, v, acc) from zero -- with the row walked in place (wave-uniform loads); one thread per pixel.
// Correctness path, not a streaming kernel.  grid = (pixel blocks, M, B).
// ---------------------------------------------------------------------------------------------
template <typename T, typename TO>
__global__ __launch_bounds__(256) void mask_mix_wide_kernel(const float *__restrict__ Rb, const T *__restrict__ masks_p,
                                                            int N, int M, int Pp, int HW, int64_t sp_b, int64_t sp_n,
                                                            const int32_t *__restrict__ n_valid,
                                                            const int32_t *__restrict__ m_valid, TO *__restrict__ out,
                                                            int64_t so_b, int64_t so_m) {
    const int b = blockIdx.z, m = blockIdx.y;
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= HW) return;
    int Nb = n_valid ? n_valid[b] : N;
    int Mb = m_valid ? m_valid[b] : M;
    if (Nb <= 0) Mb = 0;
    float acc = 0.0f;
    if (m < Mb) {
        const float *Rrow = Rb + ((int64_t)b * M + m) * Pp;
        const T *Pb = frame_base(masks_p, b, sp_b);
        for (int n = 0; n < Nb; ++n) {
            const float w = Rrow[n];
            if (w != 0.0f) acc = __builtin_fmaf(w, MaskIO<T>::load1(Pb + (int64_t)n * sp_n + x), acc);
        }
    }
    MixOut<TO>::store1(out + (int64_t)b * so_b + (int64_t)m * so_m + x, acc);
}

// dRb for ANY N, M: one workgroup per (proposal column, template row, frame); entries outside the support of Rb are zero
// without touching a plane.  Deterministic (a fixed-order block reduction, no atomics).  Correctness path.
template <typename T>
__global__ __launch_bounds__(256) void mask_mix_bwd_wide_kernel(const float *__restrict__ Rb, const T *__restrict__ masks_p,
                                                                const float *__restrict__ dout, int N, int M, int Pp,
                                                                int HW, int64_t sp_b, int64_t sp_n,
                                                                const int32_t *__restrict__ n_valid,
                                                                const int32_t *__restrict__ m_valid,
                                                                float *__restrict__ dRb) {
    __shared__ float part[4];
    const int b = blockIdx.z, m = blockIdx.y, nn = blockIdx.x;
    int Nb = n_valid ? n_valid[b] : N;
    int Mb = m_valid ? m_valid[b] : M;
    if (Nb <= 0) Mb = 0;
    float *o = dRb + ((int64_t)b * M + m) * Pp + nn;
    const bool live = m < Mb && nn < Nb && Rb[((int64_t)b * M + m) * Pp + nn] != 0.0f;
    if (!live) {
        if (threadIdx.x == 0) *o = 0.0f;
        return;
    }
    const T *plane = frame_base(masks_p, b, sp_b) + (int64_t)nn * sp_n;
    const float *drow = dout + ((int64_t)b * M + m) * HW;
    float acc = 0.0f;
    for (int x = threadIdx.x; x < HW; x += 256) acc = __builtin_fmaf(drow[x], MaskIO<T>::load1(plane + x), acc);
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *o = ((part[0] + part[1]) + part[2]) + part[3];
}

template <typename T>
static int mask_mix_bwd_wide_typed(const float *Rb, const T *masks_p, const float *dout, int B, int N, int M, int Pp, int HW,
                                   int64_t sp_b, int64_t sp_n, const int32_t *n_valid, const int32_t *m_valid, float *dRb,
                                   hipStream_t stream) {
    if (Pp > N && !g_drb_prezeroed) DMM_HIP_TRY(zero_async(dRb, sizeof(float) * (size_t)B * M * Pp, stream));   // the padded columns
    hipLaunchKernelGGL((mask_mix_bwd_wide_kernel<T>), dim3(N, M, B), dim3(256), 0, stream, Rb, masks_p, dout, N, M, Pp, HW,
                       sp_b, sp_n, n_valid, m_valid, dRb);
    return check_launch();
}

template <typename T, typename TO>
static int mask_mix_wide_typed(const float *Rb, const T *masks_p, int B, int N, int M, int Pp, int HW, int64_t sp_b,
                               int64_t sp_n, const int32_t *n_valid, const int32_t *m_valid, TO *out, int64_t so_b,
                               int64_t so_m, hipStream_t stream) {
    hipLaunchKernelGGL((mask_mix_wide_kernel<T, TO>), dim3((HW + 255) / 256, M, B), dim3(256), 0, stream, Rb, masks_p, N, M,
                       Pp, HW, sp_b, sp_n, n_valid, m_valid, out, so_b, so_m);                 // (M, B <= 65535: caller)
    return check_launch();
}

}  // namespace dmm

extern "C" int dmm_mask_mix_to(const float *Rb, const void *masks_p, int dtype, int B, int N, int M, int Pp, int HW,
                               int64_t sp_b, int64_t sp_n, const int32_t *n_valid, const int32_t *m_valid, void *out,
                               int out_dtype, int64_t so_b, int64_t so_m, dmm_stream_t stream) {
    if (B < 0 || N < 0 || M < 0 || HW < 0 || Pp < N) return DMM_ERR_BAD_ARG;
    if (B == 0 || M == 0 || HW == 0) return DMM_OK;
    if (!Rb || !masks_p || !out) return DMM_ERR_BAD_ARG;
    if (M > 65535 || B > 65535) return DMM_ERR_UNSUPPORTED;
    if (sp_n < HW || so_m < HW) return DMM_ERR_BAD_ARG;
    if (out_dtype != DMM_F32 && out_dtype != dtype) return DMM_ERR_BAD_ARG;      // fp32, or the planes' own 16-bit type
    hipStream_t s = (hipStream_t)stream;
    if (M > DMM_MAX_TEMPLATES || N > DMM_MAX_PROPOSALS || dmm::opt(DMM_OPT_FORCE_WIDE) == 1) {   // tests: the general kernel everywhere
        switch (dtype) {
            case DMM_F32:
                return dmm::mask_mix_wide_typed<float, float>(Rb, (const float *)masks_p, B, N, M, Pp, HW, sp_b, sp_n,
                                                              n_valid, m_valid, (float *)out, so_b, so_m, s);
            case DMM_F16:
                if (out_dtype == DMM_F16)
                    return dmm::mask_mix_wide_typed<dmm::f16_t, dmm::f16_t>(Rb, (const dmm::f16_t *)masks_p, B, N, M, Pp,
                                                                            HW, sp_b, sp_n, n_valid, m_valid,
                                                                            (dmm::f16_t *)out, so_b, so_m, s);
                return dmm::mask_mix_wide_typed<dmm::f16_t, float>(Rb, (const dmm::f16_t *)masks_p, B, N, M, Pp, HW, sp_b,
                                                                   sp_n, n_valid, m_valid, (float *)out, so_b, so_m, s);
            case DMM_BF16:
                if (out_dtype == DMM_BF16)
                    return dmm::mask_mix_wide_typed<dmm::bf16_t, dmm::bf16_t>(Rb, (const dmm::bf16_t *)masks_p, B, N, M,
                                                                              Pp, HW, sp_b, sp_n, n_valid, m_valid,
                                                                              (dmm::bf16_t *)out, so_b, so_m, s);
                return dmm::mask_mix_wide_typed<dmm::bf16_t, float>(Rb, (const dmm::bf16_t *)masks_p, B, N, M, Pp, HW,
                                                                    sp_b, sp_n, n_valid, m_valid, (float *)out, so_b, so_m, s);
            default:
                return DMM_ERR_BAD_ARG;
        }
    }
    // rows that share planes (train mode): DMM_OPT_MIX_SHARED -1 = the caller's entry point decides (dmm_mask_mix_shared_to),
    // 0 = row kernel always, 1 = union kernel always (tests pin both: bit-identical)
    const int shared_opt = dmm::opt(DMM_OPT_MIX_SHARED);
    if (shared_opt == 1 || (shared_opt < 0 && dmm::g_mix_shared_call)) {
        switch (dtype) {
            case DMM_F32:
                return dmm::mask_mix_shared_typed<float, float>(Rb, (const float *)masks_p, B, N, M, Pp, HW, sp_b, sp_n,
                                                                n_valid, m_valid, (float *)out, so_b, so_m, s);
            case DMM_F16:
                if (out_dtype == DMM_F16)
                    return dmm::mask_mix_shared_typed<dmm::f16_t, dmm::f16_t>(Rb, (const dmm::f16_t *)masks_p, B, N, M, Pp,
                                                                              HW, sp_b, sp_n, n_valid, m_valid,
                                                                              (dmm::f16_t *)out, so_b, so_m, s);
                return dmm::mask_mix_shared_typed<dmm::f16_t, float>(Rb, (const dmm::f16_t *)masks_p, B, N, M, Pp, HW,
                                                                     sp_b, sp_n, n_valid, m_valid, (float *)out, so_b,
                                                                     so_m, s);
            case DMM_BF16:
                if (out_dtype == DMM_BF16)
                    return dmm::mask_mix_shared_typed<dmm::bf16_t, dmm::bf16_t>(Rb, (const dmm::bf16_t *)masks_p, B, N, M,
                                                                                Pp, HW, sp_b, sp_n, n_valid, m_valid,
                                                                                (dmm::bf16_t *)out, so_b, so_m, s);
                return dmm::mask_mix_shared_typed<dmm::bf16_t, float>(Rb, (const dmm::bf16_t *)masks_p, B, N, M, Pp, HW,
                                                                      sp_b, sp_n, n_valid, m_valid, (float *)out, so_b,
                                                                      so_m, s);
            default:
                return DMM_ERR_BAD_ARG;
        }
    }
    switch (dtype) {
        case DMM_F32:
            return dmm::mask_mix_typed<float, float>(Rb, (const float *)masks_p, B, N, M, Pp, HW, sp_b, sp_n, n_valid,
                                                     m_valid, (float *)out, so_b, so_m, s);
        case DMM_F16:
            if (out_dtype == DMM_F16)
                return dmm::mask_mix_typed<dmm::f16_t, dmm::f16_t>(Rb, (const dmm::f16_t *)masks_p, B, N, M, Pp, HW, sp_b,
                                                                   sp_n, n_valid, m_valid, (dmm::f16_t *)out, so_b, so_m, s);
            return dmm::mask_mix_typed<dmm::f16_t, float>(Rb, (const dmm::f16_t *)masks_p, B, N, M, Pp, HW, sp_b, sp_n,
                                                          n_valid, m_valid, (float *)out, so_b, so_m, s);
        case DMM_BF16:
            if (out_dtype == DMM_BF16)
                return dmm::mask_mix_typed<dmm::bf16_t, dmm::bf16_t>(Rb, (const dmm::bf16_t *)masks_p, B, N, M, Pp, HW,
                                                                     sp_b, sp_n, n_valid, m_valid, (dmm::bf16_t *)out,
                                                                     so_b, so_m, s);
            return dmm::mask_mix_typed<dmm::bf16_t, float>(Rb, (const dmm::bf16_t *)masks_p, B, N, M, Pp, HW, sp_b, sp_n,
                                                           n_valid, m_valid, (float *)out, so_b, so_m, s);
        default:
            return DMM_ERR_BAD_ARG;
    }
}

// (4d) the same product for weight tables whose rows share planes: every plane of the union of the supports is streamed
// once (train mode, match_model.py:126-129,144).  Same result bit for bit.
extern "C" int dmm_mask_mix_shared_to(const float *Rb, const void *masks_p, int dtype, int B, int N, int M, int Pp, int HW,
                                      int64_t sp_b, int64_t sp_n, const int32_t *n_valid, const int32_t *m_valid,
                                      void *out, int out_dtype, int64_t so_b, int64_t so_m, dmm_stream_t stream) {
    dmm::g_mix_shared_call = true;
    const int rc = dmm_mask_mix_to(Rb, masks_p, dtype, B, N, M, Pp, HW, sp_b, sp_n, n_valid, m_valid, out, out_dtype, so_b,
                                   so_m, stream);
    dmm::g_mix_shared_call = false;
    return rc;
}

extern "C" int dmm_mask_mix_shared_frames(const float *Rb, const void *const *masks_p_frames, int dtype, int B, int N,
                                          int M, int Pp, int HW, int64_t sp_n, const int32_t *n_valid,
                                          const int32_t *m_valid, float *out, int64_t so_b, int64_t so_m,
                                          dmm_stream_t stream) {
    return dmm_mask_mix_shared_to(Rb, (const void *)masks_p_frames, dtype, B, N, M, Pp, HW, dmm::kFrameTable, sp_n, n_valid,
                                  m_valid, out, DMM_F32, so_b, so_m, stream);
}

extern "C" int dmm_mask_mix(const float *Rb, const void *masks_p, int dtype, int B, int N, int M, int Pp, int HW,
                            int64_t sp_b, int64_t sp_n, const int32_t *n_valid, const int32_t *m_valid, float *out,
                            int64_t so_b, int64_t so_m, dmm_stream_t stream) {
    return dmm_mask_mix_to(Rb, masks_p, dtype, B, N, M, Pp, HW, sp_b, sp_n, n_valid, m_valid, out, DMM_F32, so_b, so_m,
                           stream);
}

extern "C" int dmm_mask_mix_bwd(const float *Rb, const void *masks_p, int dtype, const float *dout, int B, int N, int M,
                                int Pp, int HW, int64_t sp_b, int64_t sp_n, const int32_t *n_valid,
                                const int32_t *m_valid, float *dRb, dmm_stream_t stream) {
    if (B < 0 || N < 0 || M < 0 || HW < 0 || Pp < N) return DMM_ERR_BAD_ARG;
    if (B == 0 || M == 0) return DMM_OK;
    if (!Rb || !masks_p || !dout || !dRb) return DMM_ERR_BAD_ARG;
    if (M > 65535 || B > 65535) return DMM_ERR_UNSUPPORTED;
    if (sp_n < HW) return DMM_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (M > DMM_MAX_TEMPLATES || N > DMM_MAX_PROPOSALS || dmm::opt(DMM_OPT_FORCE_WIDE) == 1) {   // any N, M: the general kernel
        switch (dtype) {
            case DMM_F32:
                return dmm::mask_mix_bwd_wide_typed<float>(Rb, (const float *)masks_p, dout, B, N, M, Pp, HW, sp_b, sp_n,
                                                           n_valid, m_valid, dRb, s);
            case DMM_F16:
                return dmm::mask_mix_bwd_wide_typed<dmm::f16_t>(Rb, (const dmm::f16_t *)masks_p, dout, B, N, M, Pp, HW, sp_b,
                                                                sp_n, n_valid, m_valid, dRb, s);
            case DMM_BF16:
                return dmm::mask_mix_bwd_wide_typed<dmm::bf16_t>(Rb, (const dmm::bf16_t *)masks_p, dout, B, N, M, Pp, HW,
                                                                 sp_b, sp_n, n_valid, m_valid, dRb, s);
            default:
                return DMM_ERR_BAD_ARG;
        }
    }
    // default: planes of the union streamed once -- while the four per-wave pair tables fit the default dynamic-LDS limit
    // (4 * N * MT floats: everything up to 112 proposals x 32 rows or 224 x 16); wider tables keep the row kernel
    const int mt = M <= 8 ? 8 : (M <= 16 ? 16 : 32);
    // The union kernel's four per-wave pair tables share ONE dynamic block with its slot slabs (mix_bwd_dynamic_lds: the
    // larger of the two) beside ~3.1 KB of static tables: 4 * N * mt floats <= 56 KB keeps the workgroup under the 64 KB a
    // launch gets without an attribute (wider tables: the row kernel).  Edge cases 112 x 32 and 224 x 16 are in the tests.
    if (dmm::opt(DMM_OPT_MIX_SHARED) != 0 && sizeof(float) * 4 * (size_t)N * mt <= 56 * 1024) {
        switch (dtype) {
            case DMM_F32:
                return dmm::mask_mix_bwd_shared_typed<float>(Rb, (const float *)masks_p, dout, B, N, M, Pp, HW, sp_b, sp_n,
                                                             n_valid, m_valid, dRb, s);
            case DMM_F16:
                return dmm::mask_mix_bwd_shared_typed<dmm::f16_t>(Rb, (const dmm::f16_t *)masks_p, dout, B, N, M, Pp, HW,
                                                                  sp_b, sp_n, n_valid, m_valid, dRb, s);
            case DMM_BF16:
                return dmm::mask_mix_bwd_shared_typed<dmm::bf16_t>(Rb, (const dmm::bf16_t *)masks_p, dout, B, N, M, Pp, HW,
                                                                   sp_b, sp_n, n_valid, m_valid, dRb, s);
            default:
                return DMM_ERR_BAD_ARG;
        }
    }
    switch (dtype) {
        case DMM_F32:
            return dmm::mask_mix_bwd_typed<float>(Rb, (const float *)masks_p, dout, B, N, M, Pp, HW, sp_b, sp_n, n_valid,
                                                  m_valid, dRb, s);
        case DMM_F16:
            return dmm::mask_mix_bwd_typed<dmm::f16_t>(Rb, (const dmm::f16_t *)masks_p, dout, B, N, M, Pp, HW, sp_b, sp_n,
                                                       n_valid, m_valid, dRb, s);
        case DMM_BF16:
            return dmm::mask_mix_bwd_typed<dmm::bf16_t>(Rb, (const dmm::bf16_t *)masks_p, dout, B, N, M, Pp, HW, sp_b,
                                                        sp_n, n_valid, m_valid, dRb, s);
        default:
            return DMM_ERR_BAD_ARG;
    }
}

// ---- per-frame pointer tables for the proposal planes (the per-video tensors of DMM_Model: no batch copy) ----
extern "C" int dmm_mask_mix_frames(const float *Rb, const void *const *masks_p_frames, int dtype, int B, int N, int M,
                                   int Pp, int HW, int64_t sp_n, const int32_t *n_valid, const int32_t *m_valid,
                                   float *out, int64_t so_b, int64_t so_m, dmm_stream_t stream) {
    return dmm_mask_mix(Rb, (const void *)masks_p_frames, dtype, B, N, M, Pp, HW, dmm::kFrameTable, sp_n, n_valid,
                        m_valid, out, so_b, so_m, stream);
}

extern "C" int dmm_mask_mix_bwd_frames(const float *Rb, const void *const *masks_p_frames, int dtype, const float *dout,
                                       int B, int N, int M, int Pp, int HW, int64_t sp_n, const int32_t *n_valid,
                                       const int32_t *m_valid, float *dRb, dmm_stream_t stream) {
    return dmm_mask_mix_bwd(Rb, (const void *)masks_p_frames, dtype, dout, B, N, M, Pp, HW, dmm::kFrameTable, sp_n,
                            n_valid, m_valid, dRb, stream);
}

// dmm_mask_mix_bwd for a caller that cleared dRb itself (dmm_match_train_backward: by the launch in front)
namespace dmm {
int mask_mix_bwd_prezeroed(const float *Rb, const void *masks_p, int dtype, const float *dout, int B, int N, int M, int Pp,
                           int HW, int64_t sp_b, int64_t sp_n, const int32_t *n_valid, const int32_t *m_valid, float *dRb,
                           dmm_stream_t stream) {
    g_drb_prezeroed = true;
    const int rc = dmm_mask_mix_bwd(Rb, masks_p, dtype, dout, B, N, M, Pp, HW, sp_b, sp_n, n_valid, m_valid, dRb, stream);
    g_drb_prezeroed = false;
    return rc;
}
}  // namespace dmm
