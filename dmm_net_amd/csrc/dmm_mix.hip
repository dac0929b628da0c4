// dmm_mix.hip -- assignment-weighted mask mix on gfx950.
//
// Replaces torch.mm(binary_Ridx_matched, pad_proposed_mask2d) of the reference
// (dmm/modules/match_model.py:134-144): full_outmask[m, :] = sum_n Rb[m, n] * mask_p[n, :].
// Rb is sparse by construction (test mode: the row maxima only; train mode: entries > 0.01), so
// only the union of proposal planes with a non-zero weight is streamed, each exactly once, and
// fanned into the <= MT output rows held in registers.
//
// Roofline: HBM.  Bytes per frame = (#selected planes + M) * HW * 4  (test mode: <= 2*M*HW*4).
#include "dmm_common.h"

namespace dmm {

constexpr int kMixThreads = 256;

constexpr int kMixUnroll = 8;    // plane loads in flight per thread

// grid = (pixel blocks, B); each thread owns 4 consecutive pixels per step.
template <typename T, int MT>
__global__ __launch_bounds__(kMixThreads) void mask_mix_kernel(const float *__restrict__ Rb, const T *__restrict__ masks_p,
                                                               int N, int M, int Pp, int HW, int64_t sp_b, int64_t sp_n,
                                                               const int32_t *__restrict__ n_valid,
                                                               const int32_t *__restrict__ m_valid,
                                                               float *__restrict__ out, int64_t so_b, int64_t so_m,
                                                               int steps_per_wg) {
    __shared__ float w_s[MT * DMM_MAX_PROPOSALS];   // compacted weights [list pos][m]
    __shared__ int col_s[DMM_MAX_PROPOSALS];        // proposal index of each list entry
    __shared__ unsigned rows_s[DMM_MAX_PROPOSALS];  // bit m set <=> weight [pos][m] != 0
    __shared__ int cnt_s;
    const int b = blockIdx.y;
    int Nb = n_valid ? n_valid[b] : N;
    int Mb = m_valid ? m_valid[b] : M;
    if (Nb <= 0) Mb = 0;
    const float *Rb_b = Rb + (int64_t)b * M * Pp;

    // Build the list of proposal planes that carry any non-zero weight (ascending n).
    if (threadIdx.x < 64) {          // one wave scans the columns in order: ballot keeps it sorted
        int base = 0;
        for (int n0 = 0; n0 < Nb; n0 += 64) {
            const int n = n0 + threadIdx.x;
            unsigned rows = 0;
            if (n < Nb)
                for (int m = 0; m < Mb; ++m) rows |= (Rb_b[(int64_t)m * Pp + n] != 0.0f) ? (1u << m) : 0u;
            const unsigned long long bal = __ballot(rows != 0);
            if (rows) {
                const int pos = base + __builtin_popcountll(bal & ((1ull << threadIdx.x) - 1ull));
                col_s[pos] = n;
                rows_s[pos] = rows;
                for (int m = 0; m < MT; ++m) w_s[pos * MT + m] = m < Mb ? Rb_b[(int64_t)m * Pp + n] : 0.0f;
            }
            base += __builtin_popcountll(bal);
        }
        if (threadIdx.x == 0) cnt_s = base;
    }
    __syncthreads();
    const int cnt = cnt_s;
    const T *Pb = masks_p + (int64_t)b * sp_b;
    float *Ob = out + (int64_t)b * so_b;

    const int nsteps = (HW + kMixThreads * 4 - 1) / (kMixThreads * 4);
    const int s_begin = blockIdx.x * steps_per_wg;
    const int s_end = min(nsteps, s_begin + steps_per_wg);
    for (int s = s_begin; s < s_end; ++s) {
        const int x = (s * kMixThreads + threadIdx.x) * 4;
        const bool in = x < HW;
        const bool full = x + 3 < HW;
        float acc[MT][4];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[m][k] = 0.0f;
        for (int e0 = 0; e0 < cnt; e0 += kMixUnroll) {
            float v[kMixUnroll][4];
#pragma unroll
            for (int u = 0; u < kMixUnroll; ++u) {
                const int e = e0 + u < cnt ? e0 + u : cnt - 1;       // clamp: extra loads hit the same plane
                const T *plane = Pb + (int64_t)col_s[e] * sp_n;
                if (full) {
                    MaskIO<T>::load4(plane + x, v[u]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[u][k] = (in && x + k < HW) ? MaskIO<T>::load1(plane + x + k) : 0.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < kMixUnroll; ++u) {
                if (e0 + u < cnt) {                                   // wave-uniform
                    const unsigned rows = rows_s[e0 + u];
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        if (rows & (1u << m)) {                       // wave-uniform
                            const float w = w_s[(e0 + u) * MT + m];
#pragma unroll
                            for (int k = 0; k < 4; ++k) acc[m][k] = __builtin_fmaf(w, v[u][k], acc[m][k]);
                        }
                    }
                }
            }
        }
        if (!in) continue;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (m < M) {                          // rows >= Mb are written as zeros
                float *o = Ob + (int64_t)m * so_m + x;
                if (full) {
                    float4u t;
                    t.x = acc[m][0]; t.y = acc[m][1]; t.z = acc[m][2]; t.w = acc[m][3];
                    *reinterpret_cast<float4u *>(o) = t;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (x + k < HW) o[k] = acc[m][k];
                }
            }
        }
    }
}

template <typename T>
static int mask_mix_typed(const float *Rb, const T *masks_p, int B, int N, int M, int Pp, int HW, int64_t sp_b,
                          int64_t sp_n, const int32_t *n_valid, const int32_t *m_valid, float *out, int64_t so_b,
                          int64_t so_m, hipStream_t stream) {
    const int nsteps = (HW + kMixThreads * 4 - 1) / (kMixThreads * 4);
    int splits = (2048 + B - 1) / B;
    if (splits > nsteps) splits = nsteps;
    if (splits < 1) splits = 1;
    const int steps_per_wg = (nsteps + splits - 1) / splits;
    splits = (nsteps + steps_per_wg - 1) / steps_per_wg;
    dim3 grid(splits, B);
#define DMM_MIX_CASE(MT_)                                                                                            \
    hipLaunchKernelGGL((mask_mix_kernel<T, MT_>), grid, dim3(kMixThreads), 0, stream, Rb, masks_p, N, M, Pp, HW, sp_b, \
                       sp_n, n_valid, m_valid, out, so_b, so_m, steps_per_wg)
    if (M <= 4) DMM_MIX_CASE(4);
    else if (M <= 8) DMM_MIX_CASE(8);
    else if (M <= 16) DMM_MIX_CASE(16);
    else DMM_MIX_CASE(32);
#undef DMM_MIX_CASE
    return check_launch();
}

}  // namespace dmm

extern "C" int dmm_mask_mix(const float *Rb, const void *masks_p, int dtype, int B, int N, int M, int Pp, int HW,
                            int64_t sp_b, int64_t sp_n, const int32_t *n_valid, const int32_t *m_valid, float *out,
                            int64_t so_b, int64_t so_m, dmm_stream_t stream) {
    if (B < 0 || N < 0 || M < 0 || HW < 0 || Pp < N) return DMM_ERR_BAD_ARG;
    if (B == 0 || M == 0 || HW == 0) return DMM_OK;
    if (!Rb || !masks_p || !out) return DMM_ERR_BAD_ARG;
    if (M > DMM_MAX_TEMPLATES || N > DMM_MAX_PROPOSALS) return DMM_ERR_UNSUPPORTED;
    if (sp_n < HW || so_m < HW) return DMM_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case DMM_F32:
            return dmm::mask_mix_typed<float>(Rb, (const float *)masks_p, B, N, M, Pp, HW, sp_b, sp_n, n_valid, m_valid,
                                              out, so_b, so_m, s);
        case DMM_F16:
            return dmm::mask_mix_typed<dmm::f16_t>(Rb, (const dmm::f16_t *)masks_p, B, N, M, Pp, HW, sp_b, sp_n, n_valid,
                                               m_valid, out, so_b, so_m, s);
        case DMM_BF16:
            return dmm::mask_mix_typed<dmm::bf16_t>(Rb, (const dmm::bf16_t *)masks_p, B, N, M, Pp, HW, sp_b, sp_n,
                                                     n_valid, m_valid, out, so_b, so_m, s);
        default:
            return DMM_ERR_BAD_ARG;
    }
}
