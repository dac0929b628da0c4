// dmm_cosine_lanes.h -- device side of the one-launch feature similarity with the D axis spread over the lanes of a wave
// (description: dmm_cosine_lanes.hip).  A header because two launches run it: cosine_lanes_kernel (dmm_cosine_lanes.hip)
// and the small-batch front kernel of dmm_match_forward (dmm_cost.hip: similarity workgroups beside the count workgroups).
#pragma once
#include "dmm_torch_order.h"

namespace dmm {

typedef float float2v __attribute__((ext_vector_type(2)));

// LDS position of element d of a row: 4 floats of padding after every 64, so that the strided reads of the class B
// lanes (d = 64 b + 4 i + j) and the 16-float block reads of the class A lanes are both conflict free.
__device__ __forceinline__ int cf_pos(int d) { return d + 4 * (d >> 6); }

// Writes of one lane become visible to the other lanes of the SAME wave: DS operations of a wave execute in issue
// order, so only the compiler has to be kept from moving them.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// a / b for many a and one b.  The fp32 division the compiler emits is
//   y0 = rcp(b'); e = fma(-b', y0, 1); y1 = fma(e, y0, y0);                       (b' = b scaled by a power of two)
//   q0 = a' y1; r0 = fma(-b', q0, a'); q1 = fma(r0, y1, q0); r1 = fma(-b', q1, a'); q = fma(r1, y1, q1)
// followed by the inverse scaling and a fix-up of the special cases.  Scaling by powers of two changes nothing while
// every intermediate stays normal; that holds when b < 2^25 and every non-zero |a| >= 2^-100 (a <= ||row|| = b keeps
// the top end in range; r0 and r1 are exact down to 2^-149), so under those two conditions the five per-element
// operations below ARE that sequence.  Anything else (tiny or non-finite values) takes the real division.
struct RowDiv {
    float b, y1;
    bool ok;
    __device__ __forceinline__ void set(float bb) {
        b = bb;
        const float y0 = __builtin_amdgcn_rcpf(bb);
        const float e = __builtin_fmaf(-bb, y0, 1.0f);
        y1 = __builtin_fmaf(e, y0, y0);
        ok = bb < 33554432.0f;                              // false for inf and NaN too
    }
    __device__ __forceinline__ float quot(float a) const {
        const float q0 = a * y1;
        const float r0 = __builtin_fmaf(-b, q0, a);
        const float q1 = __builtin_fmaf(r0, y1, q0);
        const float r1 = __builtin_fmaf(-b, q1, a);
        return __builtin_fmaf(r1, y1, q1);
    }
};
// min over (2 |a| bits - 1): zero maps to 0xFFFFFFFF, so the minimum is the smallest NON-ZERO magnitude
__device__ __forceinline__ uint32_t tiny_key(float a) { return (__float_as_uint(a) << 1) - 1u; }
constexpr uint32_t kTinyKeyMin = (((127u - 100u) << 23) << 1) - 1u;      // |a| >= 2^-100

template <int CNT>
__device__ __forceinline__ void divide_all(float (&v)[CNT], const RowDiv &rd) {
    uint32_t mn = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < CNT; ++i) { const uint32_t t = tiny_key(v[i]); mn = t < mn ? t : mn; }
    if (rd.ok && mn >= kTinyKeyMin) {
#pragma unroll
        for (int i = 0; i < CNT; ++i) v[i] = rd.quot(v[i]);
    } else {
#pragma unroll
        for (int i = 0; i < CNT; ++i) v[i] = v[i] / rd.b;
    }
}

// the 16 values of this lane's block: CA: d = 16 k + e;  class B: d = 64 (k >> 2) + 4 e + (k & 3)
template <bool CA>
__device__ __forceinline__ void load_block16(const float *row, int k, float (&v)[16]) {
    if (CA) {
        const float *p = row + cf_pos(16 * k);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 t = *reinterpret_cast<const float4 *>(p + 4 * q);
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
    } else {
        const float *p = row + cf_pos(64 * (k >> 2)) + (k & 3);
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = p[4 * e];
    }
}

template <int LPC>
struct CfGeom {
    static constexpr int D = 16 * LPC, D4 = D / 4;
    static constexpr int CPW = 64 / LPC;                    // columns side by side in a wave
    static constexpr int JB = 8 / CPW;                      // column slots per lane (8 columns per step)
    static constexpr int JH = JB / 2;                       // ... per half step (4 rows staged at a time)
    static constexpr int PITCH = D + D / 16 + 8;            // floats per LDS row: = 8 * odd (mod 64)
    static constexpr int PST = LPC + 4;                     // floats per output in the partial-sum buffer
    static constexpr int ROWBUF = 4 * PITCH;                // per-wave buffer: 4 rows, later the partial sums
    static constexpr int MG = ROWBUF / (8 * PST) < 8 ? ROWBUF / (8 * PST) : 8;   // templates per combine
    static constexpr int WAVE_FLOATS = ROWBUF + 8;
};

// one step of one wave: columns c_lo .. c_lo + w - 1 (w <= 8, all of one class) against all M templates
template <int LPC, bool CA>
__device__ __forceinline__ void cf_step(const float *__restrict__ pp, const float *Q, float *wb, float *np, int lane, int N,
                                        int M, int c_lo, int w, float *__restrict__ cos_bm) {
    typedef CfGeom<LPC> G;
    const int c = lane / LPC, k = lane % LPC;
    float2v rr[G::JB / 2][16];                              // [pair of column slots][e] = (slot 2jp, slot 2jp + 1)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        // ---- rows 4h .. 4h+3 of the step: coalesced row loads -> LDS ----
        constexpr int NLD = 4 * G::D4 / 64;
        float4u raw[NLD];
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int idx = q * 64 + lane, r = idx / G::D4, c4 = (idx - r * G::D4) * 4;
            raw[q] = (float4u){0.0f, 0.0f, 0.0f, 0.0f};
            if (4 * h + r < w) raw[q] = *reinterpret_cast<const float4u *>(pp + (int64_t)(c_lo + 4 * h + r) * G::D + c4);
        }
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int idx = q * 64 + lane, r = idx / G::D4, c4 = (idx - r * G::D4) * 4;
            *reinterpret_cast<float4 *>(wb + r * G::PITCH + cf_pos(c4)) = make_float4(raw[q].x, raw[q].y, raw[q].z, raw[q].w);
        }
        wave_lds_fence();
        // ---- norms: one aligned 8-lane group per row (lanes 32..63 repeat rows 0..3) ----
        {
            const int r = (lane >> 3) & 3;
            const float *x = wb + r * G::PITCH;
            float nr = torder::norm2_group8(G::D, lane & 7, [&](long i) { return x[cf_pos((int)i)]; });
            nr = nr > 1e-8f ? nr : 1e-8f;
            if ((lane & 7) == 0 && lane < 32) np[r] = nr;
        }
        wave_lds_fence();
        // ---- this lane's blocks, divided by the norm ----
#pragma unroll
        for (int j = 0; j < G::JH; ++j) {
            const int jb = h * G::JH + j, row = j * G::CPW + c;
            RowDiv rd;
            rd.set(np[row]);
            float v[16];
            load_block16<CA>(wb + row * G::PITCH, k, v);
            divide_all<16>(v, rd);
#pragma unroll
            for (int e = 0; e < 16; ++e) rr[jb >> 1][e][jb & 1] = v[e];
        }
        wave_lds_fence();                                   // wb is rewritten next
    }
    // ---- products: per template one partial sum per (column, block) ----
    for (int mg = 0; mg < M; mg += G::MG) {
        const int mcnt = M - mg < G::MG ? M - mg : G::MG;
        for (int mm = 0; mm < mcnt; ++mm) {
            float qv[16];
            load_block16<CA>(Q + (size_t)(mg + mm) * G::PITCH, k, qv);
#pragma unroll
            for (int jp = 0; jp < G::JB / 2; ++jp) {
                float2v a = {0.0f, 0.0f};
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float2v qq = {qv[e], qv[e]};
                    const float2v pr = qq * rr[jp][e];
                    a = a + pr;
                }
                wb[(mm * 8 + (2 * jp) * G::CPW + c) * G::PST + k] = a.x;
                wb[(mm * 8 + (2 * jp + 1) * G::CPW + c) * G::PST + k] = a.y;
            }
        }
        wave_lds_fence();
        // ---- one lane per output adds the partial sums in the cascade's order ----
        {
            const int o = lane < G::MG * 8 ? lane : G::MG * 8 - 1;
            const int mm = o >> 3, cs = o & 7;
            float p[LPC];
#pragma unroll
            for (int q = 0; q < LPC / 4; ++q) {
                const float4 t = *reinterpret_cast<const float4 *>(wb + o * G::PST + 4 * q);
                p[4 * q] = t.x; p[4 * q + 1] = t.y; p[4 * q + 2] = t.z; p[4 * q + 3] = t.w;
            }
            float res;
            if (CA) {
                torder::Cascade ca;
                ca.init(G::D);
#pragma unroll
                for (int i = 0; i < LPC; ++i) { ca.a0 = p[i]; ca.block16_done(); }
                res = ca.finish();
            } else {
                torder::Cascade c0, c1, c2, c3;
                c0.init(G::D / 4); c1.init(G::D / 4); c2.init(G::D / 4); c3.init(G::D / 4);
#pragma unroll
                for (int i = 0; i < LPC / 4; ++i) {
                    c0.a0 = p[4 * i]; c1.a0 = p[4 * i + 1]; c2.a0 = p[4 * i + 2]; c3.a0 = p[4 * i + 3];
                    c0.block16_done(); c1.block16_done(); c2.block16_done(); c3.block16_done();
                }
                res = c0.finish();
                const float p1 = c1.finish(), p2 = c2.finish(), p3 = c3.finish();
                res = res + p1;
                res = res + p2;
                res = res + p3;
            }
            if (lane < G::MG * 8 && mm < mcnt && cs < w) cos_bm[(int64_t)(mg + mm) * N + c_lo + cs] = res;
        }
        wave_lds_fence();
    }
}

// The whole similarity of frame b for the workgroup with index bx of gx along the steps axis: the first nw waves of the
// block take the steps (wave (bx, wave): bx * nw + wave, + gx * nw, ...), all of its waves stage the templates.
// lds: M * PITCH + 32 + nw * WAVE_FLOATS floats, see lanes_geom().
// ``n_valid`` (may be NULL): live proposals per frame.  ATen's order over the [D, P] slab of products depends on P -- the
// class bound of dmm_torch_order.h -- and the reference is called per frame with ITS proposals, so a ragged frame is
// reduced in the order of its own count Nb (columns from Nb on are zero filled), and Nb == 1 takes torch's contiguous
// inner reduction like cosine_kernel does.  (Round 5: the packed / fused entries used to run ragged batches through the
// dense order of the slot count, which differs from the reference's in the last bit for columns whose class changes.)
template <int LPC>
__device__ __forceinline__ void cosine_lanes_body(const float *__restrict__ feat_t, const float *__restrict__ feat_p, int N,
                                                  int M, float *__restrict__ cos_out, float *lds, int bx, int gx, int b,
                                                  int nw, const int32_t *__restrict__ n_valid = nullptr) {
    typedef CfGeom<LPC> G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *Q = lds;                                          // [M][PITCH] normalised templates
    float *nq = Q + (size_t)M * G::PITCH;                    // [32] their clamped norms
    float *wb = nq + 32 + (size_t)(wave < nw ? wave : 0) * G::WAVE_FLOATS;   // this wave's buffer
    float *np = wb + G::ROWBUF;                              // 8 floats: the norms of the staged rows
    int Nb = N;
    if (n_valid) Nb = min(max(n_valid[b], 0), N);
    float *cos_bm = cos_out + (int64_t)b * M * N;
    if (Nb < N && bx == 0)                                   // the dead columns of a ragged frame
        for (int i = threadIdx.x; i < M * (N - Nb); i += blockDim.x) {
            const int m = i / (N - Nb), j = Nb + (i - m * (N - Nb));
            cos_bm[(int64_t)m * N + j] = 0.0f;
        }
    if (Nb == 0) return;                                     // (uniform per block)
    // ---- templates: stage, norm, divide (whole block) ----
    const float *tq = feat_t + (int64_t)b * M * G::D;
    for (int i = threadIdx.x; i < M * G::D4; i += blockDim.x) {
        const int r = i / G::D4, c4 = (i - r * G::D4) * 4;
        const float4u v = *reinterpret_cast<const float4u *>(tq + (int64_t)r * G::D + c4);
        *reinterpret_cast<float4 *>(Q + r * G::PITCH + cf_pos(c4)) = make_float4(v.x, v.y, v.z, v.w);
    }
    __syncthreads();
    for (int r = threadIdx.x >> 3; r < M; r += blockDim.x >> 3) {
        const float *x = Q + r * G::PITCH;
        float nr = torder::norm2_group8(G::D, threadIdx.x & 7, [&](long i) { return x[cf_pos((int)i)]; });
        nr = nr > 1e-8f ? nr : 1e-8f;
        if ((threadIdx.x & 7) == 0) nq[r] = nr;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < M * G::D4; i += blockDim.x) {
        const int r = i / G::D4, c4 = (i - r * G::D4) * 4;
        float4 *p = reinterpret_cast<float4 *>(Q + r * G::PITCH + cf_pos(c4));
        RowDiv rd;
        rd.set(nq[r]);
        const float4 t = *p;
        float v[4] = {t.x, t.y, t.z, t.w};
        divide_all<4>(v, rd);
        *p = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
    const float *pp = feat_p + (int64_t)b * N * G::D;
    if (wave >= nw) return;
    if (Nb == 1) {
        // ONE live proposal: torch reduces the [D, 1] slab as a contiguous inner sum (cosine_kernel's Nb == 1 path): the
        // normalised row, the D products of a template and inner_sum_group8 over them -- wave 0 of the frame's first block
        if (bx != 0 || wave != 0) return;
        for (int d = lane; d < G::D; d += 64) wb[cf_pos(d)] = pp[d];
        wave_lds_fence();
        float nr = torder::norm2_group8(G::D, lane & 7, [&](long i) { return wb[cf_pos((int)i)]; });
        nr = nr > 1e-8f ? nr : 1e-8f;
        nr = __shfl(nr, 0);
        wave_lds_fence();
        for (int d = lane; d < G::D; d += 64) wb[cf_pos(d)] = wb[cf_pos(d)] / nr;
        wave_lds_fence();
        float *prod = wb + 2 * G::PITCH;                     // rows 2, 3 of the wave's buffer: D contiguous floats
        for (int m = 0; m < M; ++m) {
            for (int d = lane; d < G::D; d += 64) prod[d] = Q[(size_t)m * G::PITCH + cf_pos(d)] * wb[cf_pos(d)];
            wave_lds_fence();
            const float s = torder::inner_sum_group8(G::D, lane & 7, [&](long i) { return prod[i]; });
            if (lane == 0) cos_bm[(int64_t)m * N] = s;
            wave_lds_fence();
        }
        return;
    }
    // ---- columns: autonomous waves ----
    const int A = torder::outer_class_bound(Nb);
    const int sa = (A + 7) >> 3, sb = (Nb - A + 7) >> 3;
    for (int s = bx * nw + wave; s < sa + sb; s += gx * nw) {
        if (s < sa) {
            const int c_lo = 8 * s, w = (A - c_lo) < 8 ? (A - c_lo) : 8;
            cf_step<LPC, true>(pp, Q, wb, np, lane, N, M, c_lo, w, cos_bm);
        } else {
            const int c_lo = A + 8 * (s - sa), w = (Nb - c_lo) < 8 ? (Nb - c_lo) : 8;
            cf_step<LPC, false>(pp, Q, wb, np, lane, N, M, c_lo, w, cos_bm);
        }
    }
}

// launch geometry (host): waves per workgroup, workgroups per frame, dynamic LDS bytes; ok = false outside the envelope
struct LanesGeom {
    int nw, parts;
    size_t lds;
    bool ok;
};
template <int LPC>
inline LanesGeom lanes_geom(int B, int N, int M) {
    typedef CfGeom<LPC> G;
    const int A = N >= 8 ? 32 * (N / 32) : 4 * (N / 4);     // torder::outer_class_bound(N)
    const int S = (A + 7) / 8 + (N - A + 7) / 8;
    // few frames: one wave per SIMD and as many blocks as there are steps; many frames: one block per frame
    LanesGeom g;
    if ((long)B * ((S + 3) / 4) <= 256) { g.nw = S < 4 ? S : 4; g.parts = (S + g.nw - 1) / g.nw; }
    else {
        g.nw = S < 8 ? S : 8;
        g.parts = 512 / B;
        const int pmax = (S + g.nw - 1) / g.nw;
        g.parts = g.parts < 1 ? 1 : (g.parts > pmax ? pmax : g.parts);
    }
    g.lds = sizeof(float) * ((size_t)M * G::PITCH + 32 + (size_t)g.nw * G::WAVE_FLOATS);
    while (g.lds > 160 * 1024 - 512 && g.nw > 1) { --g.nw; g.lds -= sizeof(float) * G::WAVE_FLOATS; }
    g.ok = g.lds <= 160 * 1024 - 512;
    return g;
}

}  // namespace dmm
