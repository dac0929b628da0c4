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
// tile over ONE slab of rows, its four waves the 2 x 2 sub-tiles of 64 x 64 -- all on the same rows.  The rows of both
// operands go through LDS: 32 rows x 128 channels of dY and of X per stage, fetched with 16-byte lane loads (two per operand
// and thread) while the previous stage is being multiplied (two LDS buffers, one barrier per stage); a wave then reads its
// fragments' dwords from LDS exactly as the 64-wide kernel reads them from memory.  Every operand element now leaves L2 once
// per 128-wide tile of the other operand and in 16-byte pieces: 0.5 KB of operand per MFMA instead of 1 KB in 4-byte pieces.
// No fold: each wave stores its own sub-tile (float2 per lane: 256 contiguous bytes per 32 lanes).
template <bool PATCH>
__global__ __launch_bounds__(256) void wgrad_lds_kernel(const uint16_t *__restrict__ dy, const uint16_t *__restrict__ x,
                                                        int64_t R, int Co, int Cv, int64_t ldy, int64_t ldx,
                                                        float *__restrict__ out, int64_t group_stride, int64_t rows_per_group,
                                                        int cv_tiles, PatchGeom pg) {
    __shared__ __attribute__((aligned(16))) uint16_t sA[2][32][128];
    __shared__ __attribute__((aligned(16))) uint16_t sB[2][32][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
    // this thread's two 16-byte pieces of a stage: rows lr and lr + 16, channels 8 * lc .. 8 * lc + 7
    const int lr = threadIdx.x >> 4, lc = threadIdx.x & 15;
    u32x4w ra[2], rb[2];
    auto fetch = [&](int64_t r) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int64_t row = r + lr + 16 * q;
            const bool ok = row < r1;
            const u32x4w z = {0u, 0u, 0u, 0u};
            ra[q] = ok ? *reinterpret_cast<const u32x4w *>(dy + row * ldy + co0 + 8 * lc) : z;
            if (!PATCH) {
                rb[q] = ok ? *reinterpret_cast<const u32x4w *>(x + row * ldx + cin0 + 8 * lc) : z;
            } else {
                const int64_t qd = row / pg.Wo;
                const int pw = (int)(row - qd * pg.Wo);
                const int pb = (int)(qd / pg.Ho);
                const int ph = (int)(qd - (int64_t)pb * pg.Ho);
                const int hi = ph * pg.stride + tap_dh, wi = pw * pg.stride + tap_dw;
                const bool in = ok && hi >= 0 && hi < pg.H && wi >= 0 && wi < pg.W;
                rb[q] = in ? *reinterpret_cast<const u32x4w *>(x + (((int64_t)pb * pg.H + hi) * pg.W + wi) * ldx + cin0 + 8 * lc)
                           : z;
            }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            *reinterpret_cast<u32x4w *>(&sA[buf][lr + 16 * q][8 * lc]) = ra[q];
            *reinterpret_cast<u32x4w *>(&sB[buf][lr + 16 * q][8 * lc]) = rb[q];
        }
    };
    f32x16w acc00, acc01, acc10, acc11;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc00[k] = acc01[k] = acc10[k] = acc11[k] = 0.0f;
    if (r0 < r1) {
        fetch(r0);
        stash(0);
    }
    __syncthreads();
    int buf = 0;
    for (int64_t r = r0; r < r1; r += 32, buf ^= 1) {
        const bool more = r + 32 < r1;
        if (more) fetch(r + 32);
#pragma unroll
        for (int h = 0; h < 2; ++h) {                      // two 16-row MFMA steps per stage
            uint32_t wa[8], wb[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                wa[j] = *reinterpret_cast<const uint32_t *>(&sA[buf][16 * h + 8 * g + j][co_w + 2 * i]);
                wb[j] = *reinterpret_cast<const uint32_t *>(&sB[buf][16 * h + 8 * g + j][cv_w + 2 * i]);
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
        if (more) stash(buf ^ 1);
        __syncthreads();
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

// dw[...] = sum over the groups' partial tables, in group order (deterministic).  taps = 1: dw is [Co, Cv] like the partials;
// taps = 9: partial column v = tap * Ci + cin goes to the master's layout dw[co, cin, tap] ([Co, Ci, 3, 3] contiguous).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ part, int groups, int64_t n, int Cv,
                                                           int Ci, int taps, float *__restrict__ dw) {
    const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= n) return;
    float4 s = *reinterpret_cast<const float4 *>(part + e);
    for (int gph = 1; gph < groups; ++gph) {
        const float4 v = *reinterpret_cast<const float4 *>(part + (int64_t)gph * n + e);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (taps == 1) {
        *reinterpret_cast<float4 *>(dw + e) = s;
        return;
    }
    const int64_t co = e / Cv;
    const int v = (int)(e - co * Cv), tap = v / Ci, cin = v - tap * Ci;       // (4 consecutive v share co and tap: Ci % 64 == 0)
    float *o = dw + (co * Ci + cin) * taps + tap;
    o[0] = s.x; o[taps] = s.y; o[2 * taps] = s.z; o[3 * taps] = s.w;
}

struct WgradPlan {
    int64_t rows_per_wave, groups;       // (wide: rows per GROUP -- one workgroup takes one slab)
    int cv_tiles;
    int64_t tiles;
    bool wide;                           // the 128 x 128 LDS-staged kernel
};

static WgradPlan wgrad_plan(int64_t R, int Co, int Cv, int Ci_patch) {
    WgradPlan p;
    p.wide = !(Co & 127) && !(Cv & 127) && (Ci_patch == 0 || !(Ci_patch & 127));
    if (p.wide) {
        p.cv_tiles = Cv / 128;
        p.tiles = (int64_t)(Co / 128) * p.cv_tiles;
        // ~768 workgroups per launch, >= 2 stages of 32 rows each, partial tables of <= ~32 MB in all
        int64_t groups = (768 + p.tiles - 1) / p.tiles;
        const int64_t by_rows = (R + 63) / 64, by_bytes = (8LL << 20) / ((int64_t)Co * Cv);
        if (groups > by_rows) groups = by_rows;
        if (groups > by_bytes) groups = by_bytes;
        if (groups < 1) groups = 1;
        int64_t rpg = (R + groups - 1) / groups;
        p.rows_per_wave = (rpg + 31) / 32 * 32;
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
            hipLaunchKernelGGL((wgrad_lds_kernel<true>), grid, dim3(256), 0, stream, (const uint16_t *)dy, (const uint16_t *)x,
                               R, Co, Cv, ldy, ldx, out, n, p.rows_per_wave, p.cv_tiles, pg);
        else
            hipLaunchKernelGGL((wgrad_lds_kernel<false>), grid, dim3(256), 0, stream, (const uint16_t *)dy,
                               (const uint16_t *)x, R, Co, Cv, ldy, ldx, out, n, p.rows_per_wave, p.cv_tiles, pg);
    } else if (p.wide) {
        return DMM_ERR_UNSUPPORTED;                        // (16-byte pieces need 16-byte aligned rows)
    } else if (patch)
        hipLaunchKernelGGL((wgrad_bf16_kernel<true>), grid, dim3(256), 0, stream, (const uint16_t *)dy, (const uint16_t *)x, R,
                           Co, Cv, ldy, ldx, out, n, p.rows_per_wave, p.cv_tiles, pg);
    else
        hipLaunchKernelGGL((wgrad_bf16_kernel<false>), grid, dim3(256), 0, stream, (const uint16_t *)dy, (const uint16_t *)x,
                           R, Co, Cv, ldy, ldx, out, n, p.rows_per_wave, p.cv_tiles, pg);
    int rc = check_launch();
    if (rc != DMM_OK || direct) return rc;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, (const float *)out,
                       (int)p.groups, n, Cv, patch ? pg.Ci : Cv, patch ? 9 : 1, dw);
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
