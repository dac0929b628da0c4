// dmm_cosine.hip -- feature similarity of the matching layer on gfx950.
//
// Replaces get_cosine_score (reference dmm/utils/match_helper.py:51-64): F.cosine_similarity over the
// D axis of the expanded [O, D, P] tensors.  Under the torch (2.10) semantics the goldens were captured
// with, each vector is divided by max(||v||, 1e-8) first and the products are then summed over D:
//     cos[o, p] = sum_d (q[o,d] / qn[o]) * (k[p,d] / kn[p]).
// Both stages follow the summation order of the reference's torch CPU kernels (dmm_torch_order.h), so
// the table is bit exact against the reference -- it feeds the solver, whose exit tests are sensitive to
// the last ulp.  Work is tiny (M*N*D = 256 kFLOP per frame at the BASELINE config); the kernels are laid
// out for exactness first, parallelism across frames second.
#include "dmm_solve.h"
#include "dmm_torch_order.h"

namespace dmm {

// ---------------------------------------------------------------------------------------------
// out[r,:] = in[r,:] / max(||in[r,:]||_2, 1e-8); one aligned 8-lane group per row (8 rows per wave),
// the group plays the 8 AVX2 lanes of ATen's 2-norm fast path.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void feature_normalize_rows(const float *__restrict__ in, int64_t rows, int D,
                                                       float *__restrict__ out, float *__restrict__ norms, int64_t block) {
    const int64_t r = block * 32 + (threadIdx.x >> 3);
    const int l = threadIdx.x & 7;
    const bool valid = r < rows;
    const float *x = in + (valid ? r : 0) * D;
    float nr = torder::norm2_group8(D, l, [&](long i) { return x[i]; });
    nr = nr > 1e-8f ? nr : 1e-8f;                           // clamp_min(eps)
    nr = __shfl(nr, (threadIdx.x & 63) & ~7);               // group lane 0 -> its 8 lanes
    if (!valid) return;
    if ((D & 3) == 0) {                                     // 8 lanes x float4, 128-B coalesced per step
        for (int d = 4 * l; d < D; d += 32) {
            float4u v = *reinterpret_cast<const float4u *>(x + d);
            v.x = v.x / nr; v.y = v.y / nr; v.z = v.z / nr; v.w = v.w / nr;
            *reinterpret_cast<float4u *>(out + r * D + d) = v;
        }
    } else {
        for (int d = l; d < D; d += 8) out[r * D + d] = x[d] / nr;
    }
    if (norms && l == 0) norms[r] = nr;
}

__global__ __launch_bounds__(256) void feature_normalize_kernel(const float *__restrict__ in, int64_t rows, int D,
                                                                float *__restrict__ out, float *__restrict__ norms) {
    feature_normalize_rows(in, rows, D, out, norms, blockIdx.x);
}

// two row sets in ONE launch (the proposal and the template features of a training step's backward, which normalises
// both again instead of keeping 2 x [rows, D] alive from the forward): blocks [0, blocks_a) take set a, the rest set b
__global__ __launch_bounds__(256) void feature_normalize2_kernel(const float *__restrict__ in_a, int64_t rows_a,
                                                                 float *__restrict__ out_a, float *__restrict__ norms_a,
                                                                 const float *__restrict__ in_b, int64_t rows_b,
                                                                 float *__restrict__ out_b, float *__restrict__ norms_b,
                                                                 int D, int blocks_a, uint32_t *__restrict__ zero_ptr,
                                                                 int64_t zero_words) {
    // zero_ptr / zero_words: an unrelated table the launch clears on the side (dmm_match_train_backward: dRb, which the mix
    // backward behind it accumulates into -- saves the clearing launch in front of that kernel)
    if (zero_ptr) {
        const int64_t per = (zero_words + gridDim.x - 1) / gridDim.x, lo = (int64_t)blockIdx.x * per;
        const int64_t hi = lo + per < zero_words ? lo + per : zero_words;
        for (int64_t i = lo + threadIdx.x; i < hi; i += 256) zero_ptr[i] = 0u;
    }
    if ((int)blockIdx.x < blocks_a) feature_normalize_rows(in_a, rows_a, D, out_a, norms_a, blockIdx.x);
    else feature_normalize_rows(in_b, rows_b, D, out_b, norms_b, (int64_t)blockIdx.x - blocks_a);
}

// ---------------------------------------------------------------------------------------------
// cos[b,m,n] = sum_d RN(tn[b,m,d] * pn[b,n,d]) in the order ATen reduces the contiguous [D, N] slab of
// products over its rows: column n < outer_class_bound(N) by one cascade chain, the others by the
// ILP-4 row_sum.  grid = (M, B); thread n owns proposal column n; the proposal rows are staged through
// LDS in D-chunks (coalesced row loads, conflict-free padded column reads).  N == 1 degenerates to a
// contiguous (inner) reduction in torch; it is handled by an 8-lane group through LDS.
// ---------------------------------------------------------------------------------------------
constexpr int kCosMaxD1 = 4096;   // D limit of the N == 1 path (LDS staging)
constexpr int kCosDC = 64;        // D-chunk staged per step
constexpr int kCosLD = kCosDC + 4; // LDS row stride: 16-B aligned rows (ds_read_b128), 68 floats: N x 68 x 4 B <= 69.6 KB

// block = 256 threads = `slots` template rows of TPM = 64*ceil(N/64) threads each; grid = (ceil(M/slots), B).
// All slots of a block share the staged proposal tile.
__global__ __launch_bounds__(256, 4) void cosine_kernel(const float *__restrict__ featn_t, const float *__restrict__ featn_p,
                                                     int N, int M, int D, int tpm, const int32_t *__restrict__ n_valid,
                                                     const int32_t *__restrict__ m_valid, float *__restrict__ cos_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.y;
    const int slots = 256 / tpm;
    const int slot = threadIdx.x / tpm, n = threadIdx.x - slot * tpm;
    const int m = blockIdx.x * slots + slot;
    const int Nb = n_valid ? n_valid[b] : N;
    const int Mb = m_valid ? m_valid[b] : M;
    const bool row_live = m < Mb && Nb > 0;
    float *o = cos_out + ((int64_t)b * M + (m < M ? m : 0)) * N;
    const float *kbase = featn_p + (int64_t)b * N * D;
    const float *q = featn_t + ((int64_t)b * M + (m < M ? m : 0)) * D;
    if (Nb <= 1) {                                           // uniform per block
        if (Nb == 1) {
            float *prod = lds + slot * D;                    // [slots][D]
            if (row_live)
                for (int d = n; d < D; d += tpm) prod[d] = q[d] * kbase[d];
            __syncthreads();
            if (row_live && n < 8) {
                const float s = torder::inner_sum_group8(D, n, [&](long i) { return prod[i]; });
                if (n == 0) o[0] = s;
            }
        }
        if (m < M)
            for (int j = (row_live ? 1 : 0) + n; j < N; j += tpm) o[j] = 0.0f;
        return;
    }
    float *q_s = lds;                                        // [slots][kCosDC]
    float *tp = lds + slots * kCosDC;                        // [Nb][kCosLD]
    const bool live = row_live && n < Nb;
    const bool class_a = n < torder::outer_class_bound(Nb);
    torder::Cascade ca, c0, c1, c2, c3;
    const long g4 = D / 4;
    ca.init(D);
    c0.init(g4); c1.init(g4); c2.init(g4); c3.init(g4);
    const bool fast_ok = ca.lp == 4 && c0.lp == 4;           // level_step == 16 everywhere
    float rem[3] = {0.0f, 0.0f, 0.0f};
    for (int d0 = 0; d0 < D; d0 += kCosDC) {
        const int dc = min(kCosDC, D - d0);
        __syncthreads();
        if (dc == kCosDC && (D & 3) == 0) {                  // 16 float4 per row chunk, coalesced 256 B rows
            for (int i = threadIdx.x; i < Nb * 16; i += 256) {
                const int r = i >> 4, c4 = (i & 15) * 4;
                const float4u v = *reinterpret_cast<const float4u *>(kbase + (int64_t)r * D + d0 + c4);
                *reinterpret_cast<float4 *>(tp + r * kCosLD + c4) = make_float4(v.x, v.y, v.z, v.w);
            }
        } else {
            for (int i = threadIdx.x; i < Nb * dc; i += 256) {
                const int r = i / dc, c = i - r * dc;
                tp[r * kCosLD + c] = kbase[(int64_t)r * D + d0 + c];
            }
        }
        if (n < dc && m < M) q_s[slot * kCosDC + n] = q[d0 + n];   // tpm >= 64 >= dc
        __syncthreads();
        if (!live) continue;
        const float *row = tp + n * kCosLD;
        const float *qq = q_s + slot * kCosDC;
        if (dc == kCosDC && fast_ok) {
            if (class_a) {
#pragma unroll 1
                for (int blk = 0; blk < 4; ++blk) {
                    float a = ca.a0;
                    float4 qv[4], rv[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        qv[t] = *reinterpret_cast<const float4 *>(qq + 16 * blk + 4 * t);
                        rv[t] = *reinterpret_cast<const float4 *>(row + 16 * blk + 4 * t);
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        a = a + qv[t].x * rv[t].x;
                        a = a + qv[t].y * rv[t].y;
                        a = a + qv[t].z * rv[t].z;
                        a = a + qv[t].w * rv[t].w;
                    }
                    ca.a0 = a;
                    ca.block16_done();
                }
            } else {                                          // 64 consecutive d = 16 per ILP chain (d0 % 64 == 0)
                float a0 = c0.a0, a1 = c1.a0, a2 = c2.a0, a3 = c3.a0;
#pragma unroll 4
                for (int t = 0; t < 16; ++t) {
                    const float4 qv = *reinterpret_cast<const float4 *>(qq + 4 * t);
                    const float4 rv = *reinterpret_cast<const float4 *>(row + 4 * t);
                    a0 = a0 + qv.x * rv.x;
                    a1 = a1 + qv.y * rv.y;
                    a2 = a2 + qv.z * rv.z;
                    a3 = a3 + qv.w * rv.w;
                }
                c0.a0 = a0; c1.a0 = a1; c2.a0 = a2; c3.a0 = a3;
                c0.block16_done(); c1.block16_done(); c2.block16_done(); c3.block16_done();
            }
        } else if (class_a) {
            for (int dd = 0; dd < dc; ++dd) ca.push(qq[dd] * row[dd]);
        } else {
            for (int dd = 0; dd < dc; ++dd) {
                const long d = d0 + dd;
                const float p = qq[dd] * row[dd];
                if (d < 4 * g4) {
                    switch (d & 3) {
                        case 0: c0.push(p); break;
                        case 1: c1.push(p); break;
                        case 2: c2.push(p); break;
                        default: c3.push(p); break;
                    }
                } else {
                    rem[d - 4 * g4] = p;
                }
            }
        }
    }
    if (live) {
        float r;
        if (class_a) {
            r = ca.finish();
        } else {
            r = c0.finish();
            const float p1 = c1.finish(), p2 = c2.finish(), p3 = c3.finish();
            for (long i = 4 * g4; i < D; ++i) r = r + rem[i - 4 * g4];
            r = r + p1;
            r = r + p2;
            r = r + p3;
        }
        o[n] = r;
    }
    if (m < M) {
        if (row_live) {
            for (int j = Nb + n; j < N; j += tpm) o[j] = 0.0f;
        } else {
            for (int j = n; j < N; j += tpm) o[j] = 0.0f;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Wide tables (N > 64, dense batch, D % 64 == 0): the kernel above stages all N proposal rows once per template
// row (tpm = 128 / 256 leaves 2 / 1 template slots per block) -- 20 x per frame at N = 200, M = 20.  Here every thread
// carries RPT template rows, so a staged tile is used RPT x as often, and the RPT independent add chains hide each
// other's latency.  Same products, same ATen cascade order (level_step 16, D <= 2^19) as the fast path above.
// ---------------------------------------------------------------------------------------------
template <int RPT>
__global__ __launch_bounds__(256, 2) void cosine_rows_kernel(const float *__restrict__ featn_t,
                                                             const float *__restrict__ featn_p, int N, int M, int D,
                                                             int tpm, float *__restrict__ cos_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.y;
    const int slots = 256 / tpm;
    const int slot = threadIdx.x / tpm, n = threadIdx.x - slot * tpm;
    const int m0 = (blockIdx.x * slots + slot) * RPT;
    float *q_s = lds;                                        // [slots * RPT][kCosDC]
    float *tp = lds + slots * RPT * kCosDC;                  // [N][kCosLD]
    const float *kbase = featn_p + (int64_t)b * N * D;
    const bool live = n < N && m0 < M;
    const bool class_a = n < torder::outer_class_bound(N);
    float A[RPT][4], Bc[RPT][4][4];
#pragma unroll
    for (int r = 0; r < RPT; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            A[r][j] = 0.0f;
#pragma unroll
            for (int l = 0; l < 4; ++l) Bc[r][j][l] = 0.0f;
        }
    int cnt_a = 0, cnt_b = 0;                                // completed 16-element blocks of the class-A / class-B chains
    for (int d0 = 0; d0 < D; d0 += kCosDC) {
        __syncthreads();
        for (int i = threadIdx.x; i < N * 16; i += 256) {
            const int r = i >> 4, c4 = (i & 15) * 4;
            const float4u v = *reinterpret_cast<const float4u *>(kbase + (int64_t)r * D + d0 + c4);
            *reinterpret_cast<float4 *>(tp + r * kCosLD + c4) = make_float4(v.x, v.y, v.z, v.w);
        }
        if (n < kCosDC) {
#pragma unroll
            for (int r = 0; r < RPT; ++r)
                q_s[(slot * RPT + r) * kCosDC + n] =
                    m0 + r < M ? featn_t[((int64_t)b * M + m0 + r) * D + d0 + n] : 0.0f;
        }
        __syncthreads();
        if (!live) continue;
        const float *row = tp + n * kCosLD;
        const float *qq = q_s + slot * RPT * kCosDC;
        if (class_a) {
#pragma unroll 1
            for (int blk = 0; blk < 4; ++blk) {
                float4 rv[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) rv[t] = *reinterpret_cast<const float4 *>(row + 16 * blk + 4 * t);
#pragma unroll
                for (int r = 0; r < RPT; ++r) {
                    float a = A[r][0];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float4 qv = *reinterpret_cast<const float4 *>(qq + r * kCosDC + 16 * blk + 4 * t);
                        a = a + qv.x * rv[t].x;
                        a = a + qv.y * rv[t].y;
                        a = a + qv.z * rv[t].z;
                        a = a + qv.w * rv[t].w;
                    }
                    A[r][1] = A[r][1] + a;                   // Cascade::block16_done
                    A[r][0] = 0.0f;
                }
                ++cnt_a;
                if ((cnt_a & 15) == 0) {
#pragma unroll
                    for (int r = 0; r < RPT; ++r) { A[r][2] = A[r][2] + A[r][1]; A[r][1] = 0.0f; }
                    if ((cnt_a & 255) == 0) {
#pragma unroll
                        for (int r = 0; r < RPT; ++r) { A[r][3] = A[r][3] + A[r][2]; A[r][2] = 0.0f; }
                    }
                }
            }
        } else {                                             // ILP-4 row_sum: 16 products per chain and D-chunk
#pragma unroll 4
            for (int t = 0; t < 16; ++t) {
                const float4 rv = *reinterpret_cast<const float4 *>(row + 4 * t);
#pragma unroll
                for (int r = 0; r < RPT; ++r) {
                    const float4 qv = *reinterpret_cast<const float4 *>(qq + r * kCosDC + 4 * t);
                    Bc[r][0][0] = Bc[r][0][0] + qv.x * rv.x;
                    Bc[r][1][0] = Bc[r][1][0] + qv.y * rv.y;
                    Bc[r][2][0] = Bc[r][2][0] + qv.z * rv.z;
                    Bc[r][3][0] = Bc[r][3][0] + qv.w * rv.w;
                }
            }
            ++cnt_b;
#pragma unroll
            for (int r = 0; r < RPT; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    Bc[r][j][1] = Bc[r][j][1] + Bc[r][j][0];
                    Bc[r][j][0] = 0.0f;
                    if ((cnt_b & 15) == 0) {
                        Bc[r][j][2] = Bc[r][j][2] + Bc[r][j][1];
                        Bc[r][j][1] = 0.0f;
                        if ((cnt_b & 255) == 0) { Bc[r][j][3] = Bc[r][j][3] + Bc[r][j][2]; Bc[r][j][2] = 0.0f; }
                    }
                }
        }
    }
    if (!live) return;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        if (m0 + r >= M) break;
        float res;
        if (class_a) {
            res = A[r][0] + A[r][1];                         // Cascade::finish
            res = res + A[r][2];
            res = res + A[r][3];
        } else {
            float pj[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pj[j] = Bc[r][j][0] + Bc[r][j][1];
                pj[j] = pj[j] + Bc[r][j][2];
                pj[j] = pj[j] + Bc[r][j][3];
            }
            res = pj[0] + pj[1];                             // row_sum_scalar: (((p0 + p1) + p2) + p3), no remainder
            res = res + pj[2];
            res = res + pj[3];
        }
        cos_out[((int64_t)b * M + m0 + r) * N + n] = res;
    }
}

// ---------------------------------------------------------------------------------------------
// get_cosine_score in ONE launch (dense batches, N >= 2, D % 64 == 0).  The proposal columns are cut into tiles that
// never straddle the boundary A = outer_class_bound(N) between ATen's two reduction classes; a block takes one tile
// of one frame: it stages the tile's raw proposal rows + all M template rows into LDS ONCE with every load in flight
// at the same time ((nt + M) * (D + 4) * 4 bytes; 87 KB for the 32-column tile of the BASELINE shape), normalises
// them in place (ATen 2-norm order, clamp, IEEE division: the arithmetic of feature_normalize_kernel), and thread
// (m, j) then walks its whole D-long product chain out of LDS.  The three-launch path above stages the proposal rows
// once per D-chunk and template slot with a barrier pair per chunk: one block lives ~30 us for ~1 us of dependent
// adds.  Here the independent 16-element blocks of ATen's cascade (level_step 16) are accumulated four at a time, so
// the chain's add latency is hidden too, and every wave runs ONE class's code.  Same products, same association as
// cosine_kernel: bit identical.  grid = (tiles, B), block = roundup64(M * nt) threads.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void cosine_fused_kernel(const float *__restrict__ feat_t,
                                                           const float *__restrict__ feat_p, int N, int M, int D,
                                                           int nt, float *__restrict__ cos_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.y, nthreads = blockDim.x;
    const int A = torder::outer_class_bound(N);
    const int tiles_a = (A + nt - 1) / nt;
    const bool class_a = (int)blockIdx.x < tiles_a;
    const int c0 = class_a ? blockIdx.x * nt : A + ((int)blockIdx.x - tiles_a) * nt;
    const int c1 = min(class_a ? A : N, c0 + nt);
    const int w = c1 - c0;                                   // columns of this tile
    const int LDW = D + 4;                                   // row stride: 16-B aligned, lanes 0..7 cover all banks
    float *P = lds;                                          // [w][LDW]
    float *Q = lds + (size_t)w * LDW;                        // [M][LDW]
    float *nrm = Q + (size_t)M * LDW;                        // [w + M]
    const int rows = w + M, d4 = D >> 2;
    // ---- stage the raw rows: 8 x 16-byte loads in flight per thread and pass ----
    const int total4 = rows * d4;
    for (int base = threadIdx.x; base < total4; base += 8 * nthreads) {
        float4u v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * nthreads;
            if (i < total4) {
                const int r = i / d4, c4 = (i - r * d4) * 4;
                const float *src = r < w ? feat_p + ((int64_t)b * N + c0 + r) * D : feat_t + ((int64_t)b * M + (r - w)) * D;
                v[u] = *reinterpret_cast<const float4u *>(src + c4);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * nthreads;
            if (i < total4) {
                const int r = i / d4, c4 = (i - r * d4) * 4;
                *reinterpret_cast<float4 *>(lds + (size_t)r * LDW + c4) = make_float4(v[u].x, v[u].y, v[u].z, v[u].w);
            }
        }
    }
    __syncthreads();
    // ---- norms: one aligned 8-lane group per row (ATen's 2-norm fast path), clamp_min(eps) ----
    for (int r = threadIdx.x >> 3; r < rows; r += nthreads >> 3) {
        const float *x = lds + (size_t)r * LDW;
        float nr = torder::norm2_group8(D, threadIdx.x & 7, [&](long i) { return x[i]; });
        nr = nr > 1e-8f ? nr : 1e-8f;
        if ((threadIdx.x & 7) == 0) nrm[r] = nr;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < total4; i += nthreads) {
        const int r = i / d4, c4 = (i - r * d4) * 4;
        float4 *p = reinterpret_cast<float4 *>(lds + (size_t)r * LDW + c4);
        const float nr = nrm[r];
        float4 v = *p;
        v.x = v.x / nr; v.y = v.y / nr; v.z = v.z / nr; v.w = v.w / nr;
        *p = v;
    }
    __syncthreads();
    // ---- cos[m, c0 + j] ----
    if ((int)threadIdx.x >= M * w) return;
    const int m = threadIdx.x / w, j = threadIdx.x - m * w;
    const int n = c0 + j;
    const float *row = P + (size_t)j * LDW, *qq = Q + (size_t)m * LDW;
    float res;
    if (class_a) {                  // one cascade chain over d (multi_row_sum column)
        torder::Cascade ca;
        ca.init(D);
        for (int d0 = 0; d0 < D; d0 += 64) {                 // 4 independent 16-element blocks at a time
            float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float4 qv[4], rv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    qv[k] = *reinterpret_cast<const float4 *>(qq + d0 + 16 * k + 4 * t);
                    rv[k] = *reinterpret_cast<const float4 *>(row + d0 + 16 * k + 4 * t);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    a[k] = a[k] + qv[k].x * rv[k].x;
                    a[k] = a[k] + qv[k].y * rv[k].y;
                    a[k] = a[k] + qv[k].z * rv[k].z;
                    a[k] = a[k] + qv[k].w * rv[k].w;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) { ca.a0 = a[k]; ca.block16_done(); }
        }
        res = ca.finish();
    } else {                                                  // ILP-4 row_sum: chain j takes d = 4 k + j
        torder::Cascade c0, c1, c2, c3;
        const long g4 = D / 4;
        c0.init(g4); c1.init(g4); c2.init(g4); c3.init(g4);
        for (int d0 = 0; d0 < D; d0 += 128) {                // two 64-d segments (= one block of 16 per chain) at a time
            float a[2][4] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
            const bool two = d0 + 64 < D;
#pragma unroll 4
            for (int t = 0; t < 16; ++t) {
                const float4 q0 = *reinterpret_cast<const float4 *>(qq + d0 + 4 * t);
                const float4 r0 = *reinterpret_cast<const float4 *>(row + d0 + 4 * t);
                a[0][0] = a[0][0] + q0.x * r0.x; a[0][1] = a[0][1] + q0.y * r0.y;
                a[0][2] = a[0][2] + q0.z * r0.z; a[0][3] = a[0][3] + q0.w * r0.w;
                if (two) {
                    const float4 q1 = *reinterpret_cast<const float4 *>(qq + d0 + 64 + 4 * t);
                    const float4 r1 = *reinterpret_cast<const float4 *>(row + d0 + 64 + 4 * t);
                    a[1][0] = a[1][0] + q1.x * r1.x; a[1][1] = a[1][1] + q1.y * r1.y;
                    a[1][2] = a[1][2] + q1.z * r1.z; a[1][3] = a[1][3] + q1.w * r1.w;
                }
            }
            c0.a0 = a[0][0]; c1.a0 = a[0][1]; c2.a0 = a[0][2]; c3.a0 = a[0][3];
            c0.block16_done(); c1.block16_done(); c2.block16_done(); c3.block16_done();
            if (two) {
                c0.a0 = a[1][0]; c1.a0 = a[1][1]; c2.a0 = a[1][2]; c3.a0 = a[1][3];
                c0.block16_done(); c1.block16_done(); c2.block16_done(); c3.block16_done();
            }
        }
        res = c0.finish();
        const float p1 = c1.finish(), p2 = c2.finish(), p3 = c3.finish();
        res = res + p1;
        res = res + p2;
        res = res + p3;
    }
    cos_out[((int64_t)b * M + m) * N + n] = res;
}

}  // namespace dmm

extern "C" int dmm_feature_normalize_f32(const float *in, int64_t rows, int D, float *out, float *norms,
                                         dmm_stream_t stream) {
    if (rows < 0 || D < 0) return DMM_ERR_BAD_ARG;
    if (rows == 0 || D == 0) return DMM_OK;
    if (!in || !out) return DMM_ERR_BAD_ARG;
    const int64_t blocks = (rows + 31) / 32;
    hipLaunchKernelGGL(dmm::feature_normalize_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in,
                       rows, D, out, norms);
    return dmm::check_launch();
}

namespace dmm {
// one wave per gradient row up to this many frames (feature_sim_bwd_wave_rows_kernel).  us per call, per-frame kernel vs wave
// rows (D = 512; ~10 us of each figure is the host's): 50 x 5: B = 1 31 / 14, 64 33 / 14, 256 41 / 33, 512 56 / 61; 50 x 10:
// 46 / 14, 48 / 17, 58 / 49, 71 / 90; 200 x 20: 290 / 14, 311 / 60, 340 / 244, 413 / 482 -- every wave re-reads the rows it
// combines (L2), so the per-frame kernel's one pass wins once the chip is full
constexpr int kFeatBwdWaveMaxB = 256;
constexpr int kFeatBwdRowsMaxB = 0;      // the per-row form up to this many frames (0: never -- see dmm_feature_sim_bwd_f32)
// dmm_feature_normalize_f32 on two row sets with one launch (same kernel body: bit identical)
int feature_normalize2_launch(const float *in_a, int64_t rows_a, float *out_a, float *norms_a, const float *in_b,
                              int64_t rows_b, float *out_b, float *norms_b, int D, hipStream_t stream, void *zero_ptr,
                              size_t zero_bytes) {
    if (rows_a <= 0 || rows_b <= 0 || D <= 0 || !in_a || !in_b || !out_a || !out_b) return DMM_ERR_BAD_ARG;
    const int64_t ba = (rows_a + 31) / 32, bb = (rows_b + 31) / 32;
    if (ba + bb > 0x7fffffffLL) return DMM_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(feature_normalize2_kernel, dim3((unsigned)(ba + bb)), dim3(256), 0, stream, in_a, rows_a, out_a,
                       norms_a, in_b, rows_b, out_b, norms_b, D, (int)ba, (uint32_t *)zero_ptr, (int64_t)(zero_bytes / 4));
    return check_launch();
}
}  // namespace dmm

extern "C" int dmm_cosine_f32(const float *featn_t, const float *featn_p, int B, int N, int M, int D,
                              const int32_t *n_valid, const int32_t *m_valid, float *cos_out, dmm_stream_t stream) {
    if (B < 0 || N < 0 || M < 0 || D < 0) return DMM_ERR_BAD_ARG;
    if (B == 0 || N == 0 || M == 0) return DMM_OK;
    if (!featn_t || !featn_p || !cos_out) return DMM_ERR_BAD_ARG;
    // outside what the tiled kernels stage in LDS (N <= 256 columns, a whole row when one proposal is live): the general
    // kernel of dmm_wide.hip -- same sums, one thread per output.  DMM_OPT_FORCE_WIDE (tests) sends every call there.
    if (N > DMM_MAX_PROPOSALS || (D > dmm::kCosMaxD1 && (N == 1 || n_valid)) || dmm::opt(DMM_OPT_FORCE_WIDE) == 1)
        return dmm::launch_cosine_wide(featn_t, featn_p, B, N, M, D, n_valid, m_valid, cos_out, (hipStream_t)stream);
    const int tpm = N <= 64 ? 64 : (N <= 128 ? 128 : 256);
    const int slots = 256 / tpm;
    const int rows_min_n = dmm::opt(DMM_OPT_COS_ROWS_MIN_N);
    if (N >= rows_min_n && N > 1 && !n_valid && !m_valid && D % dmm::kCosDC == 0 && D <= (1 << 19)) {
        constexpr int RPT = 4;                               // template rows per thread
        const size_t lds4 = sizeof(float) * ((size_t)slots * RPT * dmm::kCosDC + (size_t)N * dmm::kCosLD);
        if (lds4 > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void *)dmm::cosine_rows_kernel<RPT>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
            if (e != hipSuccess) { dmm::set_last_hip_error((int)e); return DMM_ERR_LAUNCH; }
        }
        hipLaunchKernelGGL((dmm::cosine_rows_kernel<RPT>), dim3((M + slots * RPT - 1) / (slots * RPT), B), dim3(256), lds4,
                           (hipStream_t)stream, featn_t, featn_p, N, M, D, tpm, cos_out);
        return dmm::check_launch();
    }
    size_t lds = sizeof(float) * ((size_t)slots * dmm::kCosDC + (size_t)N * dmm::kCosLD);
    if (N == 1 || n_valid) {
        const size_t l1 = sizeof(float) * (size_t)D * slots;
        lds = l1 > lds ? l1 : lds;
    }
    if (lds > 160 * 1024)
        return dmm::launch_cosine_wide(featn_t, featn_p, B, N, M, D, n_valid, m_valid, cos_out, (hipStream_t)stream);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)dmm::cosine_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) { dmm::set_last_hip_error((int)e); return DMM_ERR_LAUNCH; }
    }
    hipLaunchKernelGGL(dmm::cosine_kernel, dim3((M + slots - 1) / slots, B), dim3(256), lds, (hipStream_t)stream,
                       featn_t, featn_p, N, M, D, tpm, n_valid, m_valid, cos_out);
    return dmm::check_launch();
}

namespace dmm {
int cosine_lanes_launch(const float *feat_t, const float *feat_p, int B, int N, int M, int D, float *cos_out,
                        hipStream_t stream, int32_t *zero_ptr = nullptr, int64_t zero_words = 0,
                        const int32_t *n_valid = nullptr);
}

extern "C" int dmm_cosine_features_f32(const float *feat_t, const float *feat_p, int B, int N, int M, int D,
                                       float *cos_out, dmm_stream_t stream) {
    if (B < 0 || N < 0 || M < 0 || D < 0) return DMM_ERR_BAD_ARG;
    if (B == 0 || N == 0 || M == 0) return DMM_OK;
    if (!feat_t || !feat_p || !cos_out) return DMM_ERR_BAD_ARG;
    // envelope of the one-launch form; callers fall back to normalise + normalise + cosine outside it
    if (N < 2 || N > DMM_MAX_PROPOSALS || M > DMM_MAX_TEMPLATES || D <= 0 || (D % 64) != 0 || D > (1 << 19) || B > 65535)
        return DMM_ERR_UNSUPPORTED;
    // D spread over the lanes (dmm_cosine_lanes.hip) where its envelope holds; DMM_OPT_COSINE_KERNEL = 1 keeps the
    // one-thread-per-output tile kernel below (A/B timing, and every other D)
    const bool force_tile = dmm::opt(DMM_OPT_COSINE_KERNEL) == 1;
    if (!force_tile) {
        const int rc = dmm::cosine_lanes_launch(feat_t, feat_p, B, N, M, D, cos_out, (hipStream_t)stream);
        if (rc != DMM_ERR_UNSUPPORTED) return rc;
    }
    // tile width: as many columns as fit the block (M * nt <= 1024 threads) and the LDS ((nt + M) rows of D + 4 floats)
    const size_t row_bytes = sizeof(float) * (size_t)(D + 4);
    const long lds_rows = (long)((160 * 1024 - 1024) / (row_bytes + sizeof(float))) - M;
    int nt = 1024 / M;
    if (nt > 64) nt = 64;
    if (nt > lds_rows) nt = (int)lds_rows;
    if (nt < 1) return DMM_ERR_UNSUPPORTED;
    const int A = N >= 8 ? 32 * (N / 32) : 4 * (N / 4);         // torder::outer_class_bound(N)
    // balance the tiles inside each class (e.g. A = 192, nt = 51 -> 4 tiles of 48)
    const int tiles_a = (A + nt - 1) / nt, tiles_b = (N - A + nt - 1) / nt;
    int wmax = 0;
    if (tiles_a) wmax = (A + tiles_a - 1) / tiles_a;
    if (tiles_b) { const int wb = (N - A + tiles_b - 1) / tiles_b; wmax = wb > wmax ? wb : wmax; }
    // one common width keeps the in-kernel tile arithmetic trivial: use the larger of the two balanced widths
    nt = wmax;
    const int ta = (A + nt - 1) / nt, tb = (N - A + nt - 1) / nt;
    const size_t lds = sizeof(float) * ((size_t)(nt + M) * (D + 4) + (size_t)(nt + M));
    if (lds > 160 * 1024 - 512) return DMM_ERR_UNSUPPORTED;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)dmm::cosine_fused_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { dmm::set_last_hip_error((int)e); return DMM_ERR_LAUNCH; }
    }
    const int threads = (M * nt + 63) & ~63;
    hipLaunchKernelGGL(dmm::cosine_fused_kernel, dim3(ta + tb, B), dim3(threads), lds, (hipStream_t)stream, feat_t, feat_p,
                       N, M, D, nt, cos_out);
    return dmm::check_launch();
}

namespace dmm {

// ---------------------------------------------------------------------------------------------
// Backward of the feature similarity (reference: torch autograd through get_cosine_score, match_helper.py:51-64, and
// through compute_matching_loss's mse, :48) -- one launch for what dmm_net_amd/backward.py did with ~15 eager ops:
//   dcos[m,n]  = dsim[m,n] * (1 - w)  +  d_loss * 2 (cos[m,n] - gt[m,n]) / (live entries)        (training only)
//   g_hat_p[n] = sum_m dcos[m,n] * tn[m,:]          g_hat_t[m] = sum_n dcos[m,n] * pn[n,:]
//   g_x        = g_hat / c  -  x * <g_hat, x> / (c^2 ||x||)      with c = max(||x||, eps)
// (torch clamps the norm's VALUE under no_grad and differentiates it as ||x||).  grid = (N + M, B): one block per
// feature row, 128 threads x float4.  Compared with the reference's autograd at 2e-4 relative (tree-ordered sums).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void feature_sim_bwd_kernel(
    const float *__restrict__ dsim, const float *__restrict__ cosv, const float *__restrict__ gt,
    const float *__restrict__ d_loss, float w_feat, const float *__restrict__ feat_t, const float *__restrict__ feat_p,
    const float *__restrict__ featn_t, const float *__restrict__ featn_p, const float *__restrict__ norm_t,
    const float *__restrict__ norm_p, int N, int M, int D, const int32_t *__restrict__ n_valid,
    const int32_t *__restrict__ m_valid, float *__restrict__ g_t, float *__restrict__ g_p) {
    extern __shared__ float coef[];                         // dcos slice of this row: max(N, M) floats (dynamic LDS)
    __shared__ float red[2];
    const int b = blockIdx.y, r = blockIdx.x;
    const bool is_p = r < N;
    const int row = is_p ? r : r - N;
    const int Nb = n_valid ? n_valid[b] : N, Mb = m_valid ? m_valid[b] : M;
    const int cntK = is_p ? Mb : Nb;                          // length of the contraction
    const bool live = is_p ? (row < Nb && Mb > 0) : (row < Mb && Nb > 0);
    float *g = is_p ? g_p + ((int64_t)b * N + row) * D : g_t + ((int64_t)b * M + row) * D;
    if (!live) {
        for (int d = threadIdx.x * 4; d < D; d += 512)
            for (int k = 0; k < 4 && d + k < D; ++k) g[d + k] = 0.0f;
        return;
    }
    const float lscale = (gt && d_loss) ? 2.0f * d_loss[b] / (float)(Nb * Mb) : 0.0f;
    for (int k = threadIdx.x; k < cntK; k += 128) {
        const int m = is_p ? k : row, n = is_p ? row : k;
        const int64_t idx = ((int64_t)b * M + m) * N + n;
        float c = dsim[idx] * w_feat;
        if (gt && d_loss) c += (cosv[idx] - gt[idx]) * lscale;
        coef[k] = c;
    }
    __syncthreads();
    const float *other = is_p ? featn_t + (int64_t)b * M * D : featn_p + (int64_t)b * N * D;
    const float *x = is_p ? feat_p + ((int64_t)b * N + row) * D : feat_t + ((int64_t)b * M + row) * D;
    const float c = is_p ? norm_p[(int64_t)b * N + row] : norm_t[(int64_t)b * M + row];
    float dot = 0.0f, nn = 0.0f;
    // pass 1: <g_hat, x> and ||x||^2 (D <= 512 * passes; g_hat recomputed in pass 2: it is M or N fmas per element)
    for (int d = threadIdx.x * 4; d < D; d += 512) {
        float gh[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < cntK; ++k) {
            const float ck = coef[k];
            const float *o = other + (int64_t)k * D + d;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (d + j < D) gh[j] = __builtin_fmaf(ck, o[j], gh[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (d + j < D) { dot = __builtin_fmaf(gh[j], x[d + j], dot); nn = __builtin_fmaf(x[d + j], x[d + j], nn); }
    }
    dot = wave_sum(dot);
    nn = wave_sum(nn);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = dot; }
    __syncthreads();
    const float dot_all = red[0] + red[1];
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = nn; }
    __syncthreads();
    const float nrm = __builtin_sqrtf(red[0] + red[1]);
    const float corr = nrm > 0.0f ? dot_all / (c * c * (nrm > 1e-30f ? nrm : 1e-30f)) : 0.0f;
    for (int d = threadIdx.x * 4; d < D; d += 512) {
        float gh[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < cntK; ++k) {
            const float ck = coef[k];
            const float *o = other + (int64_t)k * D + d;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (d + j < D) gh[j] = __builtin_fmaf(ck, o[j], gh[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (d + j < D) g[d + j] = gh[j] / c - x[d + j] * corr;
    }
}

// The same backward, one workgroup per FRAME with thread = feature column d (D threads, D <= 1024): the block-per-row form
// above re-reads the other side's normalised rows for every row and twice (0.55 ms per 512 frames of 50 x 10, D = 512: L2
// traffic of ~4 MB per frame for 250 KB of operands).  Here a thread keeps its column of the M normalised template rows in
// registers, walks the proposal rows once -- g_hat_p[n, d] (stored raw) and g_hat_t[m, d] (M accumulators) out of the same
// loads --, the per-row <g_hat, x> and ||x||^2 are wave sums folded over the waves in a fixed order, and a second pass over
// its own column applies the normalisation's backward.  g_hat itself is accumulated in the same order as above (fma over
// ascending k), only the two row reductions differ in order.
template <int MT>
__global__ __launch_bounds__(1024) void feature_sim_bwd_frame_kernel(
    const float *__restrict__ dsim, const float *__restrict__ cosv, const float *__restrict__ gt,
    const float *__restrict__ d_loss, float w_feat, const float *__restrict__ feat_t, const float *__restrict__ feat_p,
    const float *__restrict__ featn_t, const float *__restrict__ featn_p, const float *__restrict__ norm_t,
    const float *__restrict__ norm_p, int N, int M, int D, const int32_t *__restrict__ n_valid,
    const int32_t *__restrict__ m_valid, float *__restrict__ g_t, float *__restrict__ g_p) {
    extern __shared__ float lds[];                          // coef [M*N] | part [NW][2][N+M] | corr [N+M]
    const int b = blockIdx.x, d = threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NW = blockDim.x >> 6;
    const int Nb = n_valid ? n_valid[b] : N, Mb = m_valid ? m_valid[b] : M;
    float *coef = lds, *part = coef + M * N, *corr = part + NW * 2 * (N + M);
    float *gp_b = g_p + (int64_t)b * N * D, *gt_b = g_t + (int64_t)b * M * D;
    if (Nb <= 0 || Mb <= 0) {                               // nothing is live: every gradient row is zero
        for (int n = 0; n < N; ++n) gp_b[(int64_t)n * D + d] = 0.0f;
        for (int m = 0; m < M; ++m) gt_b[(int64_t)m * D + d] = 0.0f;
        return;
    }
    const float lscale = (gt && d_loss) ? 2.0f * d_loss[b] / (float)(Nb * Mb) : 0.0f;
    for (int e = threadIdx.x; e < M * N; e += blockDim.x) {
        const int m = e / N, n = e - m * N;
        float c = 0.0f;
        if (m < Mb && n < Nb) {
            const int64_t idx = ((int64_t)b * M + m) * N + n;
            c = dsim[idx] * w_feat;
            if (gt && d_loss) c += (cosv[idx] - gt[idx]) * lscale;
        }
        coef[e] = c;
    }
    __syncthreads();
    const float *tn_b = featn_t + (int64_t)b * M * D, *pn_b = featn_p + (int64_t)b * N * D;
    const float *xt_b = feat_t + (int64_t)b * M * D, *xp_b = feat_p + (int64_t)b * N * D;
    float tnr[MT], ght[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        tnr[m] = m < Mb ? tn_b[(int64_t)m * D + d] : 0.0f;
        ght[m] = 0.0f;
    }
    float *pw = part + wave * 2 * (N + M);
    // kFsbUn proposal rows per trip: their loads are issued together (a rolled loop paid one L2 round trip per row -- with ONE
    // frame in flight that chain was the whole 46 us of the launch) and their 2 x kFsbUn wave sums run interleaved
    constexpr int kFsbUn = 8;
    for (int n0 = 0; n0 < Nb; n0 += kFsbUn) {
        float pnv[kFsbUn], xp[kFsbUn], red[2 * kFsbUn];
#pragma unroll
        for (int u = 0; u < kFsbUn; ++u) {
            const int n = n0 + u < Nb ? n0 + u : Nb - 1;
            pnv[u] = pn_b[(int64_t)n * D + d];
            xp[u] = xp_b[(int64_t)n * D + d];
        }
#pragma unroll
        for (int u = 0; u < kFsbUn; ++u) {
            const int n = n0 + u < Nb ? n0 + u : Nb - 1;
            float ghp = 0.0f;
            if (n0 + u < Nb) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    if (m < Mb) {
                        const float c = coef[m * N + n];
                        ghp = __builtin_fmaf(c, tnr[m], ghp);
                        ght[m] = __builtin_fmaf(c, pnv[u], ght[m]);
                    }
                }
                gp_b[(int64_t)n * D + d] = ghp;             // raw g_hat; fixed up below by the same thread
            }
            red[2 * u] = ghp * xp[u];
            red[2 * u + 1] = xp[u] * xp[u];
        }
        wave_sum_rows<2 * kFsbUn>(red);
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < kFsbUn; ++u)
                if (n0 + u < Nb) { pw[n0 + u] = red[2 * u]; pw[N + M + n0 + u] = red[2 * u + 1]; }
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        if (m < Mb) {
            const float xt = xt_b[(int64_t)m * D + d];
            const float sd = wave_sum(ght[m] * xt), sq = wave_sum(xt * xt);
            if (lane == 0) { pw[N + m] = sd; pw[N + M + N + m] = sq; }
        }
    }
    __syncthreads();
    for (int r = threadIdx.x; r < N + M; r += blockDim.x) {
        const bool is_p = r < N;
        const int row = is_p ? r : r - N;
        const bool live = is_p ? row < Nb : row < Mb;
        float cr = 0.0f;
        if (live) {
            float dot = 0.0f, nn = 0.0f;
            for (int w = 0; w < NW; ++w) {
                dot += part[w * 2 * (N + M) + r];
                nn += part[w * 2 * (N + M) + N + M + r];
            }
            const float c = is_p ? norm_p[(int64_t)b * N + row] : norm_t[(int64_t)b * M + row];
            const float nrm = __builtin_sqrtf(nn);
            cr = nrm > 0.0f ? dot / (c * c * (nrm > 1e-30f ? nrm : 1e-30f)) : 0.0f;
        }
        corr[r] = cr;
    }
    __syncthreads();
    for (int n0 = 0; n0 < N; n0 += kFsbUn) {                 // the same batching for the fix-up pass
        float gh[kFsbUn], xp[kFsbUn], nr[kFsbUn];
#pragma unroll
        for (int u = 0; u < kFsbUn; ++u) {
            const int n = n0 + u < Nb ? n0 + u : Nb - 1;
            gh[u] = gp_b[(int64_t)n * D + d];
            xp[u] = xp_b[(int64_t)n * D + d];
            nr[u] = norm_p[(int64_t)b * N + n];
        }
#pragma unroll
        for (int u = 0; u < kFsbUn; ++u) {
            const int n = n0 + u;
            if (n < N) gp_b[(int64_t)n * D + d] = n < Nb ? gh[u] / nr[u] - xp[u] * corr[n] : 0.0f;
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        if (m < M) {
            float g = 0.0f;
            if (m < Mb) g = ght[m] / norm_t[(int64_t)b * M + m] - xt_b[(int64_t)m * D + d] * corr[N + m];
            gt_b[(int64_t)m * D + d] = g;
        }
    }
}


// The same backward for a HANDFUL of frames (the trainer's per-frame call, DMM_Model's 4 videos): ONE WAVE PER GRADIENT ROW.
// The per-frame kernel above walks its proposal rows in batches through three block-wide phases -- with one frame in flight
// that is ~14 dependent L2 round trips, 31 us for 50 x 5 rows of 512 features.  Here a proposal row is one wave (lane = 8
// neighbouring features of D = 512): its <= 32 coefficients sit in the lanes, the template rows it needs are all loaded at once,
// the two row sums are wave sums -- one round trip and no barrier; a template row (N terms) takes the eight waves of a
// workgroup, each summing every eighth proposal row, folded through LDS in a fixed order.  grid = (ceil(N / 8) + M, B).
// Same formulas as the kernels above; the summation order differs (inside the backward's 2e-5 bound).
template <int VEC>                                         // D == 64 * VEC
__global__ __launch_bounds__(512) void feature_sim_bwd_wave_rows_kernel(
    const float *__restrict__ dsim, const float *__restrict__ cosv, const float *__restrict__ gt,
    const float *__restrict__ d_loss, float w_feat, const float *__restrict__ feat_t, const float *__restrict__ feat_p,
    const float *__restrict__ featn_t, const float *__restrict__ featn_p, const float *__restrict__ norm_t,
    const float *__restrict__ norm_p, int N, int M, const int32_t *__restrict__ n_valid,
    const int32_t *__restrict__ m_valid, float *__restrict__ g_t, float *__restrict__ g_p) {
    constexpr int D = 64 * VEC, kRows = 8;
    __shared__ __attribute__((aligned(16))) float part[8][D];
    __shared__ float red_s[2][8];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Nb = n_valid ? n_valid[b] : N, Mb = m_valid ? m_valid[b] : M;
    const bool dead = Nb <= 0 || Mb <= 0;
    const float lscale = (!dead && gt && d_loss) ? 2.0f * d_loss[b] / (float)(Nb * Mb) : 0.0f;
    auto coef_at = [&](int m, int n) -> float {
        const int64_t idx = ((int64_t)b * M + m) * N + n;
        float c = dsim[idx] * w_feat;
        if (gt && d_loss) c += (cosv[idx] - gt[idx]) * lscale;
        return c;
    };
    auto load_row = [&](const float *row, float (&v)[VEC]) {
#pragma unroll
        for (int q = 0; q < VEC / 4; ++q) {
            const float4u t = *reinterpret_cast<const float4u *>(row + 4 * q);
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
    };
    auto store_row = [&](float *row, const float (&v)[VEC]) {
#pragma unroll
        for (int q = 0; q < VEC / 4; ++q) {
            float4u t;
            t.x = v[4 * q]; t.y = v[4 * q + 1]; t.z = v[4 * q + 2]; t.w = v[4 * q + 3];
            *reinterpret_cast<float4u *>(row + 4 * q) = t;
        }
    };
    const int pblocks = (N + 7) / 8;
    if ((int)blockIdx.x < pblocks) {
        // ---- a proposal row per wave: g_hat = sum_m coef[m, n] * tn[m];  g = g_hat / c - x * (g_hat . x) / (c^2 * |x|) ----
        const int n = blockIdx.x * 8 + wave;
        if (n >= N) return;
        float *out = g_p + ((int64_t)b * N + n) * D + lane * VEC;
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.0f;
        if (dead || n >= Nb) { store_row(out, acc); return; }
        const float cl = lane < Mb ? coef_at(lane, n) : 0.0f;              // lane m holds coef[m, n] (M <= 64)
        float x[VEC];
        load_row(feat_p + ((int64_t)b * N + n) * D + lane * VEC, x);
        const float *tn_b = featn_t + (int64_t)b * M * D + lane * VEC;
        for (int m0 = 0; m0 < Mb; m0 += kRows) {
            float t[kRows][VEC];
#pragma unroll
            for (int u = 0; u < kRows; ++u) load_row(tn_b + (int64_t)(m0 + u < Mb ? m0 + u : Mb - 1) * D, t[u]);
#pragma unroll
            for (int u = 0; u < kRows; ++u) {
                const float c = m0 + u < Mb ? __shfl(cl, m0 + u) : 0.0f;
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] = __builtin_fmaf(c, t[u][k], acc[k]);
            }
        }
        float pd = 0.0f, pq = 0.0f;
#pragma unroll
        for (int k = 0; k < VEC; ++k) { pd = __builtin_fmaf(acc[k], x[k], pd); pq = __builtin_fmaf(x[k], x[k], pq); }
        const float dot = wave_sum(pd), nn = wave_sum(pq);
        const float c = norm_p[(int64_t)b * N + n], nrm = __builtin_sqrtf(nn);
        const float cr = nrm > 0.0f ? dot / (c * c * (nrm > 1e-30f ? nrm : 1e-30f)) : 0.0f;
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = acc[k] / c - x[k] * cr;
        store_row(out, acc);
        return;
    }
    // ---- a template row per workgroup: wave w sums the proposal rows w, w + 8, ...; fixed-order fold; same fix-up ----
    const int m = blockIdx.x - pblocks;
    float *out_row = g_t + ((int64_t)b * M + m) * D;
    if (dead || m >= Mb) {
        for (int d = threadIdx.x; d < D; d += 512) out_row[d] = 0.0f;
        return;
    }
    {
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.0f;
        const int cnt = Nb > wave ? (Nb - wave + 7) / 8 : 0;              // rows of this wave: n = wave + 8 j, j < cnt (<= 64)
        const float cl = lane < cnt ? coef_at(m, wave + 8 * lane) : 0.0f; // lane j holds coef[m, wave + 8 j]
        const float *pn_b = featn_p + (int64_t)b * N * D + lane * VEC;
        for (int j0 = 0; j0 < cnt; j0 += kRows) {
            float t[kRows][VEC];
#pragma unroll
            for (int u = 0; u < kRows; ++u) {
                const int j = j0 + u < cnt ? j0 + u : cnt - 1;
                load_row(pn_b + (int64_t)(wave + 8 * j) * D, t[u]);
            }
#pragma unroll
            for (int u = 0; u < kRows; ++u) {
                const float c = j0 + u < cnt ? __shfl(cl, j0 + u) : 0.0f;
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] = __builtin_fmaf(c, t[u][k], acc[k]);
            }
        }
        store_row(&part[wave][lane * VEC], acc);
    }
    __syncthreads();
    constexpr int PER = (D + 511) / 512;
    float g[PER], xt[PER];
    float pd = 0.0f, pq = 0.0f;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int d = threadIdx.x + 512 * q;
        g[q] = 0.0f; xt[q] = 0.0f;
        if (d < D) {
#pragma unroll
            for (int w = 0; w < 8; ++w) g[q] += part[w][d];
            xt[q] = feat_t[((int64_t)b * M + m) * D + d];
            pd = __builtin_fmaf(g[q], xt[q], pd);
            pq = __builtin_fmaf(xt[q], xt[q], pq);
        }
    }
    pd = wave_sum(pd);
    pq = wave_sum(pq);
    if (lane == 0) { red_s[0][wave] = pd; red_s[1][wave] = pq; }
    __syncthreads();
    float dot = 0.0f, nn = 0.0f;
#pragma unroll
    for (int w = 0; w < 8; ++w) { dot += red_s[0][w]; nn += red_s[1][w]; }
    const float c = norm_t[(int64_t)b * M + m], nrm = __builtin_sqrtf(nn);
    const float cr = nrm > 0.0f ? dot / (c * c * (nrm > 1e-30f ? nrm : 1e-30f)) : 0.0f;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int d = threadIdx.x + 512 * q;
        if (d < D) out_row[d] = g[q] / c - xt[q] * cr;
    }
}

}  // namespace dmm

extern "C" int dmm_feature_sim_bwd_f32(const float *dsim, const float *cosv, const float *gt, const float *d_loss,
                                       float score_weight, const float *feat_t, const float *feat_p,
                                       const float *featn_t, const float *featn_p, const float *norm_t,
                                       const float *norm_p, int B, int N, int M, int D, const int32_t *n_valid,
                                       const int32_t *m_valid, float *g_feat_t, float *g_feat_p, dmm_stream_t stream) {
    if (B < 0 || N < 0 || M < 0 || D < 0) return DMM_ERR_BAD_ARG;
    if (B == 0 || N + M == 0 || D == 0) return DMM_OK;
    if (!dsim || !feat_t || !feat_p || !featn_t || !featn_p || !norm_t || !norm_p || !g_feat_t || !g_feat_p)
        return DMM_ERR_BAD_ARG;
    if ((gt || d_loss) && !(gt && d_loss && cosv)) return DMM_ERR_BAD_ARG;
    // any N, M whose longer side fits the default dynamic-LDS limit (the row's dcos slice: 4 bytes per entry)
    const size_t coef_bytes = sizeof(float) * (size_t)(N > M ? N : M);
    if (coef_bytes > 60 * 1024 || B > 65535 || (int64_t)N + M > 0x7fffffffLL) return DMM_ERR_UNSUPPORTED;
    const float w_feat = (float)(1.0 - (double)score_weight);
    // one workgroup per frame, thread = feature column, while the template rows fit a thread's registers (M <= 32) and D is
    // a whole number of waves up to one workgroup; anything else: the block-per-row form
    const int nw = D / 64;
    const size_t frame_lds = sizeof(float) * ((size_t)M * N + (size_t)nw * 2 * (N + M) + (size_t)(N + M));
    // (-1 = by batch size.  Measured at ONE frame, 50 x 5, D = 512: the per-row form 43 us, the per-frame form 46 us while its
    // row loop was rolled (one L2 round trip per proposal row), with the rows batched eight at a time see profiles/r05; at 512
    // frames the per-frame form is 5x the per-row form's throughput.  kFeatBwdRowsMaxB = 0: the per-frame form at every size)
    const int frame_mode = dmm::opt(DMM_OPT_FEAT_BWD_FRAME);
    // a handful of frames: one wave per gradient row (2 = always, -1 = up to kFeatBwdWaveMaxB frames)
    if ((frame_mode == 2 || (frame_mode < 0 && B <= dmm::kFeatBwdWaveMaxB)) && N > 0 && M > 0 && M <= 64 && N <= 512 &&
        (D == 256 || D == 512 || D == 1024)) {
#define DMM_FSW(VEC_)                                                                                                    \
    hipLaunchKernelGGL((dmm::feature_sim_bwd_wave_rows_kernel<VEC_>), dim3((N + 7) / 8 + M, B), dim3(512), 0,            \
                       (hipStream_t)stream, dsim, cosv, gt, d_loss, w_feat, feat_t, feat_p, featn_t, featn_p, norm_t,    \
                       norm_p, N, M, n_valid, m_valid, g_feat_t, g_feat_p)
        if (D == 256) DMM_FSW(4);
        else if (D == 512) DMM_FSW(8);
        else DMM_FSW(16);
#undef DMM_FSW
        return dmm::check_launch();
    }
    if ((frame_mode == 1 || (frame_mode < 0 && B > dmm::kFeatBwdRowsMaxB)) && N > 0 && M > 0 && M <= 32 && D % 64 == 0 && D <= 1024 &&
        frame_lds <= 60 * 1024) {
#define DMM_FSB(MT_)                                                                                                     \
    hipLaunchKernelGGL((dmm::feature_sim_bwd_frame_kernel<MT_>), dim3(B), dim3(D), frame_lds, (hipStream_t)stream, dsim, \
                       cosv, gt, d_loss, w_feat, feat_t, feat_p, featn_t, featn_p, norm_t, norm_p, N, M, D, n_valid,     \
                       m_valid, g_feat_t, g_feat_p)
        if (M <= 8) DMM_FSB(8);
        else if (M <= 16) DMM_FSB(16);
        else DMM_FSB(32);
#undef DMM_FSB
        return dmm::check_launch();
    }
    hipLaunchKernelGGL(dmm::feature_sim_bwd_kernel, dim3(N + M, B), dim3(128), coef_bytes, (hipStream_t)stream, dsim, cosv, gt,
                       d_loss, w_feat, feat_t, feat_p, featn_t, featn_p, norm_t, norm_p, N, M, D, n_valid, m_valid,
                       g_feat_t, g_feat_p);
    return dmm::check_launch();
}
