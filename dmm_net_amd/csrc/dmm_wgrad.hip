// dmm_wgrad.hip -- weight gradient of the encoder's convolutions on gfx950: dW = dY^T . X over the activation rows, bf16 in,
// fp32 out, MFMA (v_mfma_f32_32x32x16_bf16), split over the rows.
//
// Reference: the weight gradients autograd computes for the conv layers of the encoder under the trainer's
// loss.backward() (train.py:296-307; layers: dmm/modules/vision.py:6-38 bottlenecks, base.py:35-54 heads).  For a 1x1
// convolution on a channels-last activation that gradient IS the matrix product dW[co, ci] = sum_r dY[r, co] * X[r, ci] over
// the R = B*H*W pixels; for a 3x3 / padding 1 convolution X[r, .] is replaced by the 9 shifted pixels (virtual columns
// (kh, kw, ci): implicit im2col, nothing is materialised) and dW comes out as [co, kh, kw, ci].
//
// Why not the libraries.  The product has a tiny output (<= 2048 x 1024) and a huge reduction (R = 1 344 .. 344 064): a
// data-parallel GEMM gets a handful of tiles (hipBLASLt's pick for these shapes, MT64x64x256 without split-K: 66 us average,
// 54 launches and 3.6 ms of a 24 ms ResNet-101 step; profiles/r06_cfg4_train_encoder_steps.md), MIOpen's bf16
// split-K solvers (18-60 us) come with a zeroing and a cast launch each and clear their workspace with a memset node
// (dmm_graph.hip).  Memory bound by design: every dY / X element is read once per 64-wide tile of the other operand
// (served by L2 for the neighbouring tiles), the matrix cores idle most of the time.
//
// Kernel.  One wave = one 64 (co) x 64 (ci) tile of dW over a slab of rows; a workgroup = 4 waves = 4 consecutive slabs of
// the same tile.  Both operands are stored with the REDUCTION index (the row) as the slow dimension, while the MFMA wants 8
// consecutive k per lane.  No LDS transpose: lane (i = lane % 32, g = lane / 32) loads, for j = 0..7, the dword holding
// channels (2i, 2i+1) of row r + 8g + j; two v_perm per pair of rows turn the eight dwords into two operand fragments
// (channel 2i and channel 2i+1, each with the 8 rows as its k).  A and B use the same row-to-k assignment, so the sum over k
// is the sum over the slab's rows whatever the hardware's k order is.  4 MFMAs per 16 rows and wave, accumulators
// 4 x 16 fp32.  Epilogue WITHOUT atomics (device-scope fp32 atomics run at ~50 G/s on this part: the first form of this
// kernel, slabs x Co x Ci atomic adds, took 70-97 us per launch -- profiles/r06_cfg4_train_encoder_steps.md): the four
// waves of a workgroup are folded through LDS in a fixed order, the workgroup stores its [64, 64] tile into the partial table
// of its slab group, and a second launch sums the groups in order (and, for 3x3, writes the master's [co, ci, kh, kw]
// layout).  Deterministic; dW is overwritten, never pre-zeroed.  Slabs: ~2048 waves per launch, partial tables <= 16 MB.
#include "dmm_common.h"

namespace dmm {

#ifndef WGRAD_DEPTH
#define WGRAD_DEPTH 2
#endif
#ifndef WGRAD_KS
#define WGRAD_KS 2
#endif
typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));
typedef float f32x16w __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4w __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pair_lo(uint32_t w0, uint32_t w1) { return __builtin_amdgcn_perm(w1, w0, 0x05040100u); }
__device__ __forceinline__ uint32_t pair_hi(uint32_t w0, uint32_t w1) { return __builtin_amdgcn_perm(w1, w0, 0x07060302u); }

struct PatchGeom {       // 3x3 / padding 1 convolution: rows of dY are output pixels (b, ho, wo); X is [B, H, W, Ci]
    int H, W, Ho, Wo, stride, Ci;
};

// PATCH = false: X is [R, ldx] and virtual column c is column c.  PATCH = true: virtual column c = tap * Ci + cin.
// out: this workgroup's [Co, Cv] fp32 table = out + blockIdx.y * group_stride (the partial table of its group of 4 slabs;
// the gradient itself when the launch has one group).  Plain stores: no zeroing, no atomics.
template <bool PATCH>
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(const uint16_t *__restrict__ dy, const uint16_t *__restrict__ x,
                                                         int64_t R, int Co, int Cv, int64_t ldy, int64_t ldx,
                                                         float *__restrict__ out, int64_t group_stride,
                                                         int64_t rows_per_wave, int cv_tiles, PatchGeom pg) {
    __shared__ __attribute__((aligned(16))) float tile_s[64 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, g = lane >> 5;
    const int tile = blockIdx.x;
    const int co0 = (tile / cv_tiles) * 64, cv0 = (tile % cv_tiles) * 64;
    const int64_t slab = (int64_t)blockIdx.y * 4 + wave;
    const int64_t r0 = slab * rows_per_wave;
    int64_t r1 = r0 + rows_per_wave;
    if (r1 > R) r1 = R;
    int tap_dh = 0, tap_dw = 0, cin0 = cv0;
    if (PATCH) {
        const int tap = cv0 / pg.Ci;
        cin0 = cv0 - tap * pg.Ci;
        tap_dh = tap / 3 - 1;
        tap_dw = tap % 3 - 1;
    }
    const uint16_t *ap = dy + co0 + 2 * i;
    const uint16_t *bp = x + cin0 + 2 * i;
    f32x16w acc00, acc01, acc10, acc11;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc00[k] = acc01[k] = acc10[k] = acc11[k] = 0.0f;
    for (int64_t r = r0; r < r1; r += 16) {              // (a wave whose slab starts beyond R runs no step: zeros)
        uint32_t wa[8], wb[8];
        const int64_t rb = r + 8 * g;
        int pb = 0, ph = 0, pw = 0;
        if (PATCH) {                                       // (b, ho, wo) of the lane's first row, then incremented
            const int64_t q = rb / pg.Wo;
            pw = (int)(rb - q * pg.Wo);
            pb = (int)(q / pg.Ho);
            ph = (int)(q - (int64_t)pb * pg.Ho);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t row = rb + j;
            const bool ok = row < r1;
            wa[j] = ok ? *reinterpret_cast<const uint32_t *>(ap + row * ldy) : 0u;
            if (!PATCH) {
                wb[j] = ok ? *reinterpret_cast<const uint32_t *>(bp + row * ldx) : 0u;
            } else {
                const int hi = ph * pg.stride + tap_dh, wi = pw * pg.stride + tap_dw;
                const bool in = ok && hi >= 0 && hi < pg.H && wi >= 0 && wi < pg.W;
                wb[j] = in ? *reinterpret_cast<const uint32_t *>(bp + (((int64_t)pb * pg.H + hi) * pg.W + wi) * ldx) : 0u;
                if (++pw == pg.Wo) {
                    pw = 0;
                    if (++ph == pg.Ho) { ph = 0; ++pb; }
                }
            }
        }
        u32x4w a0, a1, b0, b1;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            a0[p] = pair_lo(wa[2 * p], wa[2 * p + 1]);
            a1[p] = pair_hi(wa[2 * p], wa[2 * p + 1]);
            b0[p] = pair_lo(wb[2 * p], wb[2 * p + 1]);
            b1[p] = pair_hi(wb[2 * p], wb[2 * p + 1]);
        }
        const bf16x8w fa0 = __builtin_bit_cast(bf16x8w, a0), fa1 = __builtin_bit_cast(bf16x8w, a1);
        const bf16x8w fb0 = __builtin_bit_cast(bf16x8w, b0), fb1 = __builtin_bit_cast(bf16x8w, b1);
        acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb0, acc00, 0, 0, 0);
        acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb1, acc01, 0, 0, 0);
        acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb0, acc10, 0, 0, 0);
        acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb1, acc11, 0, 0, 0);
    }
    // fold the four waves (four slabs of the same tile) in a FIXED order through one 16 KB LDS tile.
    // D layout: column (n, the B side) = lane & 31, row (m, the A side) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    // tile row = 2 m + c, tile column = 2 i + d
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int m = (reg & 3) + 8 * (reg >> 2) + 4 * g;
                float2 *t0 = reinterpret_cast<float2 *>(&tile_s[(2 * m) * 64 + 2 * i]);
                float2 *t1 = reinterpret_cast<float2 *>(&tile_s[(2 * m + 1) * 64 + 2 * i]);
                float2 v0 = make_float2(acc00[reg], acc01[reg]), v1 = make_float2(acc10[reg], acc11[reg]);
                if (w > 0) {
                    const float2 p0 = *t0, p1 = *t1;
                    v0.x += p0.x; v0.y += p0.y; v1.x += p1.x; v1.y += p1.y;
                }
                *t0 = v0;
                *t1 = v1;
            }
        }
        __syncthreads();
    }
    float *o = out + (int64_t)blockIdx.y * group_stride;
    const int row = threadIdx.x >> 2, c4 = (threadIdx.x & 3) * 16;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        *reinterpret_cast<float4 *>(o + (int64_t)(co0 + row) * Cv + cv0 + c4 + 4 * k) =
            *reinterpret_cast<const float4 *>(&tile_s[row * 64 + c4 + 4 * k]);
}

// The same product for shapes whose widths are multiples of 128 (everything from layer2 on): one workgroup = a 128 (co) x 128 (cv)
// tile over ONE slab of rows.  The rows of both operands go through LDS in stages of 32 rows x 128 channels of dY and of X,
// fetched with 16-byte lane loads; a wave reads its fragments' dwords from LDS exactly as the 64-wide kernel reads them from
// memory.  Every operand element leaves L2 once per 128-wide tile of the other operand and in 16-byte pieces: 0.5 KB of operand
// per MFMA instead of 1 KB in 4-byte pieces.
//
// KS k-slices.  A stage is a dependent chain (LDS store -> barrier -> LDS reads -> v_perm -> MFMA): measured 0.68 us per stage
// and workgroup whether one or two workgroups share a compute unit -- latency, not throughput (time = 2.5 + 0.68 * stages +
// 0.24 * MB of partial table, tools/wgrad_probe.py).  So a workgroup is KS x 4 waves: slice s multiplies stages s, s + KS, ...
// of the slab into its own accumulators (own LDS buffers, the 2 x 2 sub-tiles of 64 x 64 over its four waves), the slices
// are folded through LDS in a fixed order at the end.  KS x the waves per compute unit for the same bytes of partial table --
// which is what more slabs would cost.  D register sets keep D stages in flight between memory and LDS.
//
// Branch-free inner loop: a piece outside the slab / the image is loaded from a valid address and zeroed when it moves to LDS
// (loads under a branch make the compiler wait for ALL outstanding loads before every LDS store), and stages are fetched in
// order, so row offsets and the (image, row, column) of a piece's output pixel advance by additions and two carries (the four
// 64-bit divisions per stage of the first form were most of its ~700 instructions: the 3x3 form was VALU bound).
template <bool PATCH, int KS>
__global__ __launch_bounds__(256 * KS) void wgrad_lds_kernel(const uint16_t *__restrict__ dy, const uint16_t *__restrict__ x,
                                                             int64_t R, int Co, int Cv, int64_t ldy, int64_t ldx,
                                                             float *__restrict__ out, int64_t group_stride,
                                                             int64_t rows_per_group, int cv_tiles, PatchGeom pg) {
    constexpr int D = WGRAD_DEPTH;
    __shared__ __attribute__((aligned(16))) uint16_t lds_[KS * 2 * 2 * 32 * 128];        // [slice][buffer][A | B][32][128]
    const int slice = threadIdx.x >> 8, t = threadIdx.x & 255;
    const int lane = t & 63, wave = t >> 6;
    const int i = lane & 31, g = lane >> 5;
    const int tile = blockIdx.x;
    const int co0 = (tile / cv_tiles) * 128, cv0 = (tile % cv_tiles) * 128;
    const int co_w = (wave >> 1) * 64, cv_w = (wave & 1) * 64;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_group;
    int64_t r1 = r0 + rows_per_group;
    if (r1 > R) r1 = R;
    int tap_dh = 0, tap_dw = 0, cin0 = cv0;
    if (PATCH) {
        const int tap = cv0 / pg.Ci;
        cin0 = cv0 - tap * pg.Ci;
        tap_dh = tap / 3 - 1;
        tap_dw = tap % 3 - 1;
    }
    uint16_t *sA0 = lds_ + slice * (2 * 2 * 32 * 128), *sB0 = sA0 + 32 * 128;             // buffer b: + b * 2 * 32 * 128
    // this thread's two 16-byte pieces of a stage: rows lr and lr + 16, channels 8 * lc .. 8 * lc + 7
    const int lr = t >> 4, lc = t & 15;
    u32x4w ra[D][2], rb[D][2];
    uint32_t okm[D];                                      // bit q: row of piece q inside the slab; bit 2 + q: its X pixel inside the image
    int64_t a_off[2], b_off[2], row_n[2];
    int pb_[2] = {0, 0}, ph_[2] = {0, 0}, pw_[2] = {0, 0};
    int adv_w = 0, adv_h = 0, adv_b = 0;
    const int64_t a_first = r0 * ldy + co0 + 8 * lc, b_first = PATCH ? 0 : r0 * ldx + cin0 + 8 * lc;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        row_n[q] = r0 + 32 * slice + lr + 16 * q;
        a_off[q] = row_n[q] * ldy + co0 + 8 * lc;
        b_off[q] = row_n[q] * ldx + cin0 + 8 * lc;
        if (PATCH) {
            const int64_t qd = row_n[q] / pg.Wo;
            pw_[q] = (int)(row_n[q] - qd * pg.Wo);
            pb_[q] = (int)(qd / pg.Ho);
            ph_[q] = (int)(qd - (int64_t)pb_[q] * pg.Ho);
        }
    }
    if (PATCH) {                                          // 32 * KS output pixels further = adv_b images + adv_h rows + adv_w columns
        adv_w = (32 * KS) % pg.Wo;
        const int q_ = (32 * KS) / pg.Wo;
        adv_h = q_ % pg.Ho;
        adv_b = q_ / pg.Ho;
    }
    auto fetch = [&](u32x4w (&fa)[2], u32x4w (&fb)[2], uint32_t &ok_bits) {
        ok_bits = 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const bool ok = row_n[q] < r1;
            fa[q] = *reinterpret_cast<const u32x4w *>(dy + (ok ? a_off[q] : a_first));
            if (!PATCH) {
                fb[q] = *reinterpret_cast<const u32x4w *>(x + (ok ? b_off[q] : b_first));
                ok_bits |= ok ? (5u << q) : 0u;
                b_off[q] += 32 * KS * ldx;
            } else {
                const int hi = ph_[q] * pg.stride + tap_dh, wi = pw_[q] * pg.stride + tap_dw;
                const bool in = ok && hi >= 0 && hi < pg.H && wi >= 0 && wi < pg.W;
                const int64_t off = in ? (((int64_t)pb_[q] * pg.H + hi) * pg.W + wi) * ldx : 0;
                fb[q] = *reinterpret_cast<const u32x4w *>(x + off + cin0 + 8 * lc);
                ok_bits |= (ok ? (1u << q) : 0u) | (in ? (4u << q) : 0u);
                pw_[q] += adv_w;
                const int cw = pw_[q] >= pg.Wo;
                pw_[q] -= cw ? pg.Wo : 0;
                ph_[q] += adv_h + cw;
                const int ch = ph_[q] >= pg.Ho;
                ph_[q] -= ch ? pg.Ho : 0;
                pb_[q] += adv_b + ch;
            }
            a_off[q] += 32 * KS * ldy;
            row_n[q] += 32 * KS;
        }
    };
    auto stash = [&](int buf, const u32x4w (&fa)[2], const u32x4w (&fb)[2], uint32_t ok_bits) {
        const u32x4w z = {0u, 0u, 0u, 0u};
        uint16_t *sa = sA0 + buf * (2 * 32 * 128), *sb = sB0 + buf * (2 * 32 * 128);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            *reinterpret_cast<u32x4w *>(sa + (lr + 16 * q) * 128 + 8 * lc) = (ok_bits >> q) & 1u ? fa[q] : z;
            *reinterpret_cast<u32x4w *>(sb + (lr + 16 * q) * 128 + 8 * lc) = (ok_bits >> (2 + q)) & 1u ? fb[q] : z;
        }
    };
    f32x16w acc00, acc01, acc10, acc11;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc00[k] = acc01[k] = acc10[k] = acc11[k] = 0.0f;
    // stage: registers -> LDS buffer, the registers refilled D stages ahead, ONE barrier, multiply.  (The buffer the next stage
    // overwrites was read by the stage before this one; this stage's barrier lies between.)  The stage count is padded to a
    // multiple of D -- the launcher makes slabs of 32 * KS * D rows, so only the last slab multiplies a few stages of zeros --:
    // the loop body is ONE straight-line block and the compiler counts the outstanding loads exactly.
#pragma unroll
    for (int d = 0; d < D; ++d) fetch(ra[d], rb[d], okm[d]);
    const int64_t rounds = (r1 - r0 + 32 * KS * D - 1) / (32 * KS * D);
    for (int64_t it = 0; it < rounds; ++it) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int buf = (d & 1) ^ ((D & 1) ? (int)(it & 1) : 0);
            stash(buf, ra[d], rb[d], okm[d]);
            fetch(ra[d], rb[d], okm[d]);
            __syncthreads();
            const uint16_t *sa = sA0 + buf * (2 * 32 * 128), *sb = sB0 + buf * (2 * 32 * 128);
#pragma unroll
            for (int h = 0; h < 2; ++h) {                  // two 16-row MFMA steps per stage
                uint32_t wa[8], wb[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    wa[j] = *reinterpret_cast<const uint32_t *>(sa + (16 * h + 8 * g + j) * 128 + co_w + 2 * i);
                    wb[j] = *reinterpret_cast<const uint32_t *>(sb + (16 * h + 8 * g + j) * 128 + cv_w + 2 * i);
                }
                u32x4w a0, a1, b0, b1;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    a0[p] = pair_lo(wa[2 * p], wa[2 * p + 1]);
                    a1[p] = pair_hi(wa[2 * p], wa[2 * p + 1]);
                    b0[p] = pair_lo(wb[2 * p], wb[2 * p + 1]);
                    b1[p] = pair_hi(wb[2 * p], wb[2 * p + 1]);
                }
                const bf16x8w fa0 = __builtin_bit_cast(bf16x8w, a0), fa1 = __builtin_bit_cast(bf16x8w, a1);
                const bf16x8w fb0 = __builtin_bit_cast(bf16x8w, b0), fb1 = __builtin_bit_cast(bf16x8w, b1);
                acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb0, acc00, 0, 0, 0);
                acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb1, acc01, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb0, acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb1, acc11, 0, 0, 0);
            }
        }
    }
    // fold the slices, a fixed tree: slices [h, 2h) hand their 64 accumulators per thread to slices [0, h) through LDS
    // (region per pair: [64][256] floats = the operand buffers of two slices)
    if (KS > 1) {
        float *ex = reinterpret_cast<float *>(lds_);
#pragma unroll
        for (int h = KS / 2; h >= 1; h /= 2) {
            __syncthreads();                               // (the last stage's LDS reads / the previous round's are done)
            if (slice >= h && slice < 2 * h) {
                float *e = ex + (slice - h) * (64 * 256) + t;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    e[(k) * 256] = acc00[k];
                    e[(16 + k) * 256] = acc01[k];
                    e[(32 + k) * 256] = acc10[k];
                    e[(48 + k) * 256] = acc11[k];
                }
            }
            __syncthreads();
            if (slice < h) {
                const float *e = ex + slice * (64 * 256) + t;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    acc00[k] += e[(k) * 256];
                    acc01[k] += e[(16 + k) * 256];
                    acc10[k] += e[(32 + k) * 256];
                    acc11[k] += e[(48 + k) * 256];
                }
            }
        }
        if (slice != 0) return;
    }
    float *o = out + (int64_t)blockIdx.y * group_stride;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int m = (reg & 3) + 8 * (reg >> 2) + 4 * g;
        float *p0 = o + (int64_t)(co0 + co_w + 2 * m) * Cv + cv0 + cv_w + 2 * i;
        *reinterpret_cast<float2 *>(p0) = make_float2(acc00[reg], acc01[reg]);
        *reinterpret_cast<float2 *>(p0 + Cv) = make_float2(acc10[reg], acc11[reg]);
    }
}

// The LDS form for the NARROW layers (layer1 of a ResNet: 64 output or 64 input channels), tile TCO x TCV in {64 x 128, 128 x 64}:
// the 64-wide kernel above feeds its MFMAs with 4-byte lane loads (16 load instructions per 4 MFMAs: 22 / 56 us per launch on
// layer1's shapes, a quarter of the step's weight-gradient time) where this one stages 16-byte pieces through LDS like
// wgrad_lds_kernel.  A slice = the tile's 64 x 64 sub-tiles, one wave each (2 here), over its own stages; KS slices; the stage
// loop, the branch-free fetch and the fold are wgrad_lds_kernel's.  A 128-wide cv tile may hold TWO taps of a 3x3 patch
// (Ci = 64) -- the tap belongs to the thread's piece column -- and may end beyond Cv (9 x 64 = 4.5 tiles): such pieces are zeros
// and their sub-tile is not stored.
template <bool PATCH, int KS, int TCO, int TCV>
__global__ __launch_bounds__(64 * (TCO / 64) * (TCV / 64) * KS) void wgrad_tile_kernel(
    const uint16_t *__restrict__ dy, const uint16_t *__restrict__ x, int64_t R, int Co, int Cv, int64_t ldy, int64_t ldx,
    float *__restrict__ out, int64_t group_stride, int64_t rows_per_group, int cv_tiles, PatchGeom pg) {
    constexpr int D = 2;
    constexpr int WV = (TCO / 64) * (TCV / 64), TS = 64 * WV;            // waves / threads of a slice
    constexpr int PRA = TCO / 8, PRB = TCV / 8;                           // 16-byte pieces per row
    constexpr int NA = 32 * PRA / TS, NB = 32 * PRB / TS;                 // pieces per thread and stage
    constexpr int RSA = TS / PRA, RSB = TS / PRB;                         // rows between a thread's pieces
    constexpr int SLICE = 2 * 32 * (TCO + TCV);                           // ushorts of a slice's two buffers
    static_assert(TS % PRA == 0 && TS % PRB == 0 && NA >= 1 && NB >= 1, "piece mapping");
    static_assert(KS * SLICE * 2 >= (KS / 2) * 64 * TS * 4, "fold regions");
    __shared__ __attribute__((aligned(16))) uint16_t lds_[KS * SLICE];
    const int slice = threadIdx.x / TS, t = threadIdx.x % TS;
    const int lane = t & 63, wave = t >> 6;
    const int i = lane & 31, g = lane >> 5;
    const int tile = blockIdx.x;
    const int co0 = (tile / cv_tiles) * TCO, cv0 = (tile % cv_tiles) * TCV;
    const int co_w = (wave / (TCV / 64)) * 64, cv_w = (wave % (TCV / 64)) * 64;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_group;
    int64_t r1 = r0 + rows_per_group;
    if (r1 > R) r1 = R;
    const int ca = t % PRA, ra0 = t / PRA, cb = t % PRB, rb0 = t / PRB;
    const int vb = cv0 + 8 * cb;                           // the B pieces' virtual column
    const bool colb = vb < Cv;
    int tap_dh = 0, tap_dw = 0, cinb = vb;
    if (PATCH) {
        const int tap = (colb ? vb : 0) / pg.Ci;
        cinb = (colb ? vb : 0) - tap * pg.Ci;
        tap_dh = tap / 3 - 1;
        tap_dw = tap % 3 - 1;
    } else if (!colb) {
        cinb = 0;
    }
    uint16_t *sl = lds_ + slice * SLICE;                   // buffer b: A at sl + b * 32 * (TCO + TCV), B behind it
    u32x4w ra[D][NA], rb[D][NB];
    uint32_t okm[D];                                      // bit q: A piece q inside the slab; bit 8 + q: B piece q inside slab and image
    int64_t a_off[NA], rowa[NA], b_off[NB], rowb[NB];
    int pb_[NB], ph_[NB], pw_[NB];
    int adv_w = 0, adv_h = 0, adv_b = 0;
    const int64_t a_first = r0 * ldy + co0 + 8 * ca, b_first = PATCH ? cinb : r0 * ldx + cinb;
#pragma unroll
    for (int q = 0; q < NA; ++q) {
        rowa[q] = r0 + 32 * slice + ra0 + RSA * q;
        a_off[q] = rowa[q] * ldy + co0 + 8 * ca;
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        rowb[q] = r0 + 32 * slice + rb0 + RSB * q;
        b_off[q] = rowb[q] * ldx + cinb;
        pb_[q] = ph_[q] = pw_[q] = 0;
        if (PATCH) {
            const int64_t qd = rowb[q] / pg.Wo;
            pw_[q] = (int)(rowb[q] - qd * pg.Wo);
            pb_[q] = (int)(qd / pg.Ho);
            ph_[q] = (int)(qd - (int64_t)pb_[q] * pg.Ho);
        }
    }
    if (PATCH) {
        adv_w = (32 * KS) % pg.Wo;
        const int q_ = (32 * KS) / pg.Wo;
        adv_h = q_ % pg.Ho;
        adv_b = q_ / pg.Ho;
    }
    auto fetch = [&](u32x4w (&fa)[NA], u32x4w (&fb)[NB], uint32_t &ok_bits) {
        ok_bits = 0;
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const bool ok = rowa[q] < r1;
            fa[q] = *reinterpret_cast<const u32x4w *>(dy + (ok ? a_off[q] : a_first));
            ok_bits |= ok ? (1u << q) : 0u;
            a_off[q] += 32 * KS * ldy;
            rowa[q] += 32 * KS;
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const bool ok = rowb[q] < r1 && colb;
            if (!PATCH) {
                fb[q] = *reinterpret_cast<const u32x4w *>(x + (ok ? b_off[q] : b_first));
                ok_bits |= ok ? (0x100u << q) : 0u;
                b_off[q] += 32 * KS * ldx;
            } else {
                const int hi = ph_[q] * pg.stride + tap_dh, wi = pw_[q] * pg.stride + tap_dw;
                const bool in = ok && hi >= 0 && hi < pg.H && wi >= 0 && wi < pg.W;
                const int64_t off = in ? (((int64_t)pb_[q] * pg.H + hi) * pg.W + wi) * ldx : 0;
                fb[q] = *reinterpret_cast<const u32x4w *>(x + off + cinb);
                ok_bits |= in ? (0x100u << q) : 0u;
                pw_[q] += adv_w;
                const int cw = pw_[q] >= pg.Wo;
                pw_[q] -= cw ? pg.Wo : 0;
                ph_[q] += adv_h + cw;
                const int ch = ph_[q] >= pg.Ho;
                ph_[q] -= ch ? pg.Ho : 0;
                pb_[q] += adv_b + ch;
            }
            rowb[q] += 32 * KS;
        }
    };
    auto stash = [&](int buf, const u32x4w (&fa)[NA], const u32x4w (&fb)[NB], uint32_t ok_bits) {
        const u32x4w z = {0u, 0u, 0u, 0u};
        uint16_t *sa = sl + buf * (32 * (TCO + TCV)), *sb = sa + 32 * TCO;
#pragma unroll
        for (int q = 0; q < NA; ++q)
            *reinterpret_cast<u32x4w *>(sa + (ra0 + RSA * q) * TCO + 8 * ca) = (ok_bits >> q) & 1u ? fa[q] : z;
#pragma unroll
        for (int q = 0; q < NB; ++q)
            *reinterpret_cast<u32x4w *>(sb + (rb0 + RSB * q) * TCV + 8 * cb) = (ok_bits >> (8 + q)) & 1u ? fb[q] : z;
    };
    f32x16w acc00, acc01, acc10, acc11;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc00[k] = acc01[k] = acc10[k] = acc11[k] = 0.0f;
#pragma unroll
    for (int d = 0; d < D; ++d) fetch(ra[d], rb[d], okm[d]);
    const int64_t rounds = (r1 - r0 + 32 * KS * D - 1) / (32 * KS * D);
    for (int64_t it = 0; it < rounds; ++it) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int buf = d & 1;                         // (D = 2)
            stash(buf, ra[d], rb[d], okm[d]);
            fetch(ra[d], rb[d], okm[d]);
            __syncthreads();
            const uint16_t *sa = sl + buf * (32 * (TCO + TCV)), *sb = sa + 32 * TCO;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint32_t wa[8], wb[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    wa[j] = *reinterpret_cast<const uint32_t *>(sa + (16 * h + 8 * g + j) * TCO + co_w + 2 * i);
                    wb[j] = *reinterpret_cast<const uint32_t *>(sb + (16 * h + 8 * g + j) * TCV + cv_w + 2 * i);
                }
                u32x4w a0, a1, b0, b1;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    a0[p] = pair_lo(wa[2 * p], wa[2 * p + 1]);
                    a1[p] = pair_hi(wa[2 * p], wa[2 * p + 1]);
                    b0[p] = pair_lo(wb[2 * p], wb[2 * p + 1]);
                    b1[p] = pair_hi(wb[2 * p], wb[2 * p + 1]);
                }
                const bf16x8w fa0 = __builtin_bit_cast(bf16x8w, a0), fa1 = __builtin_bit_cast(bf16x8w, a1);
                const bf16x8w fb0 = __builtin_bit_cast(bf16x8w, b0), fb1 = __builtin_bit_cast(bf16x8w, b1);
                acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb0, acc00, 0, 0, 0);
                acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb1, acc01, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb0, acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb1, acc11, 0, 0, 0);
            }
        }
    }
    if (KS > 1) {
        float *ex = reinterpret_cast<float *>(lds_);
#pragma unroll
        for (int h = KS / 2; h >= 1; h /= 2) {
            __syncthreads();
            if (slice >= h && slice < 2 * h) {
                float *e = ex + (slice - h) * (64 * TS) + t;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    e[(k) * TS] = acc00[k];
                    e[(16 + k) * TS] = acc01[k];
                    e[(32 + k) * TS] = acc10[k];
                    e[(48 + k) * TS] = acc11[k];
                }
            }
            __syncthreads();
            if (slice < h) {
                const float *e = ex + slice * (64 * TS) + t;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    acc00[k] += e[(k) * TS];
                    acc01[k] += e[(16 + k) * TS];
                    acc10[k] += e[(32 + k) * TS];
                    acc11[k] += e[(48 + k) * TS];
                }
            }
        }
        if (slice != 0) return;
    }
    if (cv0 + cv_w >= Cv) return;                          // (a sub-tile beyond the last virtual column)
    float *o = out + (int64_t)blockIdx.y * group_stride;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int m = (reg & 3) + 8 * (reg >> 2) + 4 * g;
        float *p0 = o + (int64_t)(co0 + co_w + 2 * m) * Cv + cv0 + cv_w + 2 * i;
        *reinterpret_cast<float2 *>(p0) = make_float2(acc00[reg], acc01[reg]);
        *reinterpret_cast<float2 *>(p0 + Cv) = make_float2(acc10[reg], acc11[reg]);
    }
}

// dw[...] = sum over the groups' partial tables (a fixed order: deterministic).  T9 = false: dw is [Co, Cv] like the partials, a
// thread owns 4 consecutive outputs.  T9 = true: partial column v = tap * Ci + cin goes to the master's layout dw[co, cin, tap]
// ([Co, Ci, 3, 3] contiguous); a thread owns the 9 taps of one (co, cin), the workgroup writes its 256 / S x 9 contiguous outputs
// through LDS.  S = threads per output: the groups are dealt round robin to S slices which are folded through LDS -- with one
// thread per output a [64, 256] gradient summed over 123 groups was 16 workgroups of 123 dependent loads: 30 us.
template <int S, bool T9>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ part, int groups, int64_t n, int Cv,
                                                           int Ci, float *__restrict__ dw) {
    constexpr int OUT = 256 / S, W = T9 ? 9 : 4;
    __shared__ float fold[S][OUT * W + 1];
    const int o = threadIdx.x % OUT, sl = threadIdx.x / OUT;
    const int64_t idx = (int64_t)blockIdx.x * OUT + o;                      // float4 index / (co, cin) index
    const int64_t total = T9 ? n / 9 : n / 4;
    float acc[W];
#pragma unroll
    for (int k = 0; k < W; ++k) acc[k] = 0.0f;
    if (idx < total) {
        if (!T9) {
            for (int gph = sl; gph < groups; gph += S) {
                const float4 v = *reinterpret_cast<const float4 *>(part + (int64_t)gph * n + idx * 4);
                acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
            }
        } else {
            const int64_t co = idx / Ci;
            const int cin = (int)(idx - co * Ci);
            const float *p = part + co * Cv + cin;
            for (int gph = sl; gph < groups; gph += S) {
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) acc[tp] += p[(int64_t)gph * n + (int64_t)tp * Ci];
            }
        }
    }
    if (S == 1 && !T9) {
        if (idx < total) *reinterpret_cast<float4 *>(dw + idx * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        return;
    }
#pragma unroll
    for (int k = 0; k < W; ++k) fold[sl][o * W + k] = acc[k];
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * OUT * W;                     // the workgroup's outputs are contiguous in dw
    for (int j = threadIdx.x; j < OUT * W; j += 256) {
        if (base + j >= n) break;
        float v = fold[0][j];
#pragma unroll
        for (int q = 1; q < S; ++q) v += fold[q][j];
        dw[base + j] = v;
    }
}

struct WgradPlan {
    int64_t rows_per_wave, groups;       // (wide / narrow LDS forms: rows per GROUP -- one workgroup takes one slab)
    int cv_tiles;
    int64_t tiles;
    int narrow;                          // 0 none, 1 = 64 (co) x 128 (cv) tiles, 2 = 128 x 64 (wgrad_tile_kernel)
    bool wide;                           // the 128 x 128 LDS-staged kernel
};

static WgradPlan wgrad_plan(int64_t R, int Co, int Cv, int Ci_patch) {
    WgradPlan p;
    p.wide = !(Co & 127) && !(Cv & 127) && (Ci_patch == 0 || !(Ci_patch & 127));
    p.narrow = 0;
    if (!p.wide && Co == 64 && Cv >= 128 && !(Cv & 63)) p.narrow = 1;
    else if (!p.wide && Cv == 64 && !(Co & 127) && Ci_patch == 0) p.narrow = 2;
    if (p.narrow) {
        const int tco = p.narrow == 1 ? 64 : 128, tcv = p.narrow == 1 ? 128 : 64;
        constexpr int64_t RND = 32 * 4 * 2;               // rows x slices x depth of wgrad_tile_kernel
        p.cv_tiles = (Cv + tcv - 1) / tcv;
        p.tiles = (int64_t)(Co / tco) * p.cv_tiles;
        int64_t groups = 256 / p.tiles;                   // one workgroup (8 waves, 96 KB of LDS) per compute unit: ONE wave of them
        const int64_t by_rows = (R + RND - 1) / RND, by_bytes = (4LL << 20) / ((int64_t)Co * Cv);
        if (groups > by_rows) groups = by_rows;
        if (groups > by_bytes) groups = by_bytes;
        if (groups < 1) groups = 1;
        const int64_t rpg = (R + groups - 1) / groups;
        p.rows_per_wave = (rpg + RND - 1) / RND * RND;
        p.groups = (R + p.rows_per_wave - 1) / p.rows_per_wave;
        return p;
    }
    if (p.wide) {
        p.cv_tiles = Cv / 128;
        p.tiles = (int64_t)(Co / 128) * p.cv_tiles;
        // ~512 workgroups per launch, partial tables of <= ~16 MB in all, slabs of whole rounds (32 rows x slices x depth):
        // the best of a sweep over {256, 512, 768} x {8, 16, 32} MB x KS {1, 2, 4} x D {1, 2, 3} on ResNet-101's shapes at
        // 12 x 255 x 448 (tools/wgrad_probe.py: 1.84 ms for a step's 109 gradients; 2.76 with 768 / 32 MB / KS 1 / D 1)
        constexpr int64_t RND = 32 * WGRAD_KS * WGRAD_DEPTH;
        int64_t groups = (512 + p.tiles - 1) / p.tiles;
        const int64_t by_rows = (R + RND - 1) / RND, by_bytes = (4LL << 20) / ((int64_t)Co * Cv);
        if (groups > by_rows) groups = by_rows;
        if (groups > by_bytes) groups = by_bytes;
        if (groups < 1) groups = 1;
        int64_t rpg = (R + groups - 1) / groups;
        p.rows_per_wave = (rpg + RND - 1) / RND * RND;
        p.groups = (R + p.rows_per_wave - 1) / p.rows_per_wave;
        return p;
    }
    p.cv_tiles = Cv / 64;
    p.tiles = (int64_t)(Co / 64) * p.cv_tiles;
    // slabs: ~2048 waves per launch, >= 8 steps of 16 rows per wave, partial tables of <= ~16 MB in all; 4 slabs = 1 group
    int64_t slabs = (2048 + p.tiles - 1) / p.tiles;
    const int64_t by_rows = (R + 127) / 128;
    if (slabs > by_rows) slabs = by_rows;
    int64_t groups = (slabs + 3) / 4;
    const int64_t by_bytes = (4LL << 20) / ((int64_t)Co * Cv);
    if (groups > by_bytes) groups = by_bytes;
    if (groups < 1) groups = 1;
    int64_t rpw = (R + groups * 4 - 1) / (groups * 4);
    p.rows_per_wave = (rpw + 15) / 16 * 16;
    p.groups = (R + p.rows_per_wave * 4 - 1) / (p.rows_per_wave * 4);
    return p;
}

static int wgrad_launch(const void *dy, const void *x, int64_t R, int Co, int Cv, int64_t ldy, int64_t ldx, float *dw,
                        void *workspace, size_t workspace_bytes, bool patch, PatchGeom pg, hipStream_t stream) {
    const WgradPlan p = wgrad_plan(R, Co, Cv, patch ? pg.Ci : 0);
    if (p.tiles > 0x7fffffffLL || p.groups > 65535) return DMM_ERR_UNSUPPORTED;
    const int64_t n = (int64_t)Co * Cv;
    const bool direct = p.groups == 1 && !patch;          // one group and nothing to transpose: straight into dw
    if (!direct && (!workspace || workspace_bytes < sizeof(float) * (size_t)(p.groups * n))) return DMM_ERR_WORKSPACE;
    float *out = direct ? dw : (float *)workspace;
    const dim3 grid((unsigned)p.tiles, (unsigned)p.groups);
    if (p.wide && (ldy & 7) == 0 && (ldx & 7) == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)x & 15) == 0) {
        if (patch)
            hipLaunchKernelGGL((wgrad_lds_kernel<true, WGRAD_KS>), grid, dim3(256 * WGRAD_KS), 0, stream, (const uint16_t *)dy,
                               (const uint16_t *)x, R, Co, Cv, ldy, ldx, out, n, p.rows_per_wave, p.cv_tiles, pg);
        else
            hipLaunchKernelGGL((wgrad_lds_kernel<false, WGRAD_KS>), grid, dim3(256 * WGRAD_KS), 0, stream, (const uint16_t *)dy,
                               (const uint16_t *)x, R, Co, Cv, ldy, ldx, out, n, p.rows_per_wave, p.cv_tiles, pg);
    } else if (p.wide) {
        return DMM_ERR_UNSUPPORTED;                        // (16-byte pieces need 16-byte aligned rows)
    } else if (p.narrow && (ldy & 7) == 0 && (ldx & 7) == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)x & 15) == 0) {
        if (p.narrow == 1 && patch)
            hipLaunchKernelGGL((wgrad_tile_kernel<true, 4, 64, 128>), grid, dim3(512), 0, stream, (const uint16_t *)dy,
                               (const uint16_t *)x, R, Co, Cv, ldy, ldx, out, n, p.rows_per_wave, p.cv_tiles, pg);
        else if (p.narrow == 1)
            hipLaunchKernelGGL((wgrad_tile_kernel<false, 4, 64, 128>), grid, dim3(512), 0, stream, (const uint16_t *)dy,
                               (const uint16_t *)x, R, Co, Cv, ldy, ldx, out, n, p.rows_per_wave, p.cv_tiles, pg);
        else
            hipLaunchKernelGGL((wgrad_tile_kernel<false, 4, 128, 64>), grid, dim3(512), 0, stream, (const uint16_t *)dy,
                               (const uint16_t *)x, R, Co, Cv, ldy, ldx, out, n, p.rows_per_wave, p.cv_tiles, pg);
    } else if (p.narrow) {
        return DMM_ERR_UNSUPPORTED;
    } else if (patch)
        hipLaunchKernelGGL((wgrad_bf16_kernel<true>), grid, dim3(256), 0, stream, (const uint16_t *)dy, (const uint16_t *)x, R,
                           Co, Cv, ldy, ldx, out, n, p.rows_per_wave, p.cv_tiles, pg);
    else
        hipLaunchKernelGGL((wgrad_bf16_kernel<false>), grid, dim3(256), 0, stream, (const uint16_t *)dy, (const uint16_t *)x,
                           R, Co, Cv, ldy, ldx, out, n, p.rows_per_wave, p.cv_tiles, pg);
    int rc = check_launch();
    if (rc != DMM_OK || direct) return rc;
    // threads per output: enough workgroups to fill the chip (>= ~1024 where the table allows), never more slices than groups
    const int64_t outs = patch ? n / 9 : n / 4;
    int S = 1;
    while (S < 16 && 2 * S <= p.groups && outs * S < 256 * 1024) S *= 2;
    const int ci_ = patch ? pg.Ci : Cv;
#define DMM_WRED(S_)                                                                                                          \
    do {                                                                                                                      \
        const dim3 rg((unsigned)((outs + 256 / S_ - 1) / (256 / S_)));                                                        \
        if (patch)                                                                                                            \
            hipLaunchKernelGGL((wgrad_reduce_kernel<S_, true>), rg, dim3(256), 0, stream, (const float *)out, (int)p.groups,  \
                               n, Cv, ci_, dw);                                                                               \
        else                                                                                                                  \
            hipLaunchKernelGGL((wgrad_reduce_kernel<S_, false>), rg, dim3(256), 0, stream, (const float *)out, (int)p.groups, \
                               n, Cv, ci_, dw);                                                                               \
    } while (0)
    switch (S) {
        case 1: DMM_WRED(1); break;
        case 2: DMM_WRED(2); break;
        case 4: DMM_WRED(4); break;
        case 8: DMM_WRED(8); break;
        default: DMM_WRED(16); break;
    }
#undef DMM_WRED
    return check_launch();
}

}  // namespace dmm

extern "C" size_t dmm_wgrad_workspace_bytes(int64_t rows, int co, int cv) {
    if (rows <= 0 || co <= 0 || cv <= 0 || (co & 63) || (cv & 63)) return 0;
    // (the 3x3 form asks with cv = 9 * ci: 128 divides 9 * ci exactly when it divides ci, so the plan is the launcher's)
    const dmm::WgradPlan p = dmm::wgrad_plan(rows, co, cv, 0);
    return sizeof(float) * (size_t)(p.groups * (int64_t)co * cv);
}

extern "C" int dmm_wgrad_bf16(const void *dy, const void *x, int64_t rows, int co, int ci, int64_t ldy, int64_t ldx, float *dw,
                              void *workspace, size_t workspace_bytes, dmm_stream_t stream) {
    if (rows < 0 || co <= 0 || ci <= 0 || ldy < co || ldx < ci) return DMM_ERR_BAD_ARG;
    if (!dw) return DMM_ERR_BAD_ARG;
    if ((co & 63) || (ci & 63) || (ldy & 1) || (ldx & 1)) return DMM_ERR_UNSUPPORTED;
    if (rows == 0) {
        DMM_HIP_TRY(dmm::zero_async(dw, sizeof(float) * (size_t)co * ci, (hipStream_t)stream));
        return DMM_OK;
    }
    if (!dy || !x) return DMM_ERR_BAD_ARG;
    return dmm::wgrad_launch(dy, x, rows, co, ci, ldy, ldx, dw, workspace, workspace_bytes, false, dmm::PatchGeom{},
                             (hipStream_t)stream);
}

extern "C" int dmm_wgrad3x3_bf16(const void *dy, const void *x, int B, int H, int W, int ci, int co, int stride, float *dw,
                                 void *workspace, size_t workspace_bytes, dmm_stream_t stream) {
    if (B < 0 || H <= 0 || W <= 0 || co <= 0 || ci <= 0 || (stride != 1 && stride != 2)) return DMM_ERR_BAD_ARG;
    if (!dw) return DMM_ERR_BAD_ARG;
    if ((co & 63) || (ci & 63)) return DMM_ERR_UNSUPPORTED;
    if (B == 0) {
        DMM_HIP_TRY(dmm::zero_async(dw, sizeof(float) * (size_t)co * 9 * ci, (hipStream_t)stream));
        return DMM_OK;
    }
    if (!dy || !x) return DMM_ERR_BAD_ARG;
    dmm::PatchGeom pg;
    pg.H = H; pg.W = W; pg.stride = stride; pg.Ci = ci;
    pg.Ho = (H - 1) / stride + 1;
    pg.Wo = (W - 1) / stride + 1;
    const int64_t R = (int64_t)B * pg.Ho * pg.Wo;
    return dmm::wgrad_launch(dy, x, R, co, 9 * ci, co, ci, dw, workspace, workspace_bytes, true, pg, (hipStream_t)stream);
}
