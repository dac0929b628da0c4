// dmm_roialign.hip -- fused 4-level ROIAlign + spatial mean on gfx950 (reference a9).
//
// Replaces the reference's ROI feature extractor (dmm/modules/feature_extractor.py:11-52): for each of
// the 4 pyramid levels (scales 1/4 .. 1/32, :13) maskrcnn_benchmark's legacy (non-aligned) ROIAlign
// (14x14 bins, sampling_ratio 2, :14-16) is run on EVERY roi (:50-51), the [R, 4, C, 14, 14] result is
// materialised (:49) and then averaged over the 14x14 bins (:29) -> [R, 4*C].
// Third-party arithmetic: maskrcnn_benchmark (github.com/ZENGXH/maskrcnn-benchmark, un-pinned HEAD,
// INSTALL.md:24) is not vendored; its published ROIAlign definition is restated in oracle/dmm_oracle.c
// (its roialign4_mean restatement); no reference fixture exists for it -- forward and gradients are pinned against
// G12, an independent differentiable formulation of the published per-bin definition (tests/golden/gen_golden.py).
//
// The 2x2 samples of the 14x14 bins form a uniform 28x28 grid over the roi and bilinear weights are
// separable, so   out[r, l, c] = sum_h wy[h] * sum_w wx[w] * feat_l[b, c, h, w]   with two 1-D weight
// vectors per (roi, level).  The 20 MB/frame intermediate of the reference is never formed; every feature
// cell inside the roi is read once per channel.
//
// Roofline: L2 bandwidth (feature maps are a few MB per frame and are re-read by overlapping rois).
#include "dmm_common.h"

namespace dmm {

constexpr int kRoiMaxDim = 1024;     // max H or W of a feature map
constexpr int kRoiSamples = 28;      // 14 bins x sampling_ratio 2
constexpr int kPatchMax = 1024;      // cells of a roi patch flattened into LDS per tile (29 x 29 fits)
constexpr int kRoiCG = 4;            // channels a wave runs together

struct RoiLevels {
    const void *feat[4];
    float *dfeat[4];
    int H[4], W[4];
    float scale[4];
};

// 1-D weights of the 28 sample points of a roi along one axis (legacy ROIAlign bilinear rule).
__device__ __forceinline__ void axis_weights(float start, float end, int size, float *w_s, int *lo_out, int *hi_out) {
    float len = end - start;
    len = len > 1.0f ? len : 1.0f;                       // roi size clamped to >= 1 (legacy, non-aligned)
    const float bin = len / 14.0f;
    int lo = size, hi = -1;
    for (int k = 0; k < kRoiSamples; ++k) {
        const int p = k >> 1, i = k & 1;
        float y = start + (float)p * bin + ((float)i + 0.5f) * bin / 2.0f;
        if (y < -1.0f || y > (float)size) continue;      // sample outside: contributes 0
        if (y <= 0.0f) y = 0.0f;
        int y_low = (int)y, y_high;
        if (y_low >= size - 1) { y_high = y_low = size - 1; y = (float)y_low; }
        else y_high = y_low + 1;
        const float ly = y - (float)y_low, hy = 1.0f - ly;
        w_s[y_low] += hy;
        w_s[y_high] += ly;
        lo = y_low < lo ? y_low : lo;
        hi = y_high > hi ? y_high : hi;
    }
    *lo_out = lo;
    *hi_out = hi;
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<f16_t>(f16_t v) { return (float)v.v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return __uint_as_float(((uint32_t)v.v) << 16); }

// grid = (R, 4); block = 256 (4 waves, each takes every 4th channel; lanes run along w).
// BWD = false: out[r, l*C + c] = weighted sum.  BWD = true: dfeat[b,c,h,w] += wy*wx*dout[r, l*C + c] (fp32 atomics).
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void roialign4_mean_kernel(RoiLevels lv, int B, int C, const float *__restrict__ rois,
                                                             int R, float *__restrict__ out_or_dout) {
    __shared__ float wy_s[kRoiMaxDim], wx_s[kRoiMaxDim];
    __shared__ float wgt_s[BWD ? 1 : kPatchMax];
    __shared__ int off_s[BWD ? 1 : kPatchMax];
    __shared__ int rng_s[4];
    const int r = blockIdx.x, l = blockIdx.y;
    const int H = lv.H[l], W = lv.W[l];
    const float sc = lv.scale[l];
    for (int i = threadIdx.x; i < H; i += 256) wy_s[i] = 0.0f;
    for (int i = threadIdx.x; i < W; i += 256) wx_s[i] = 0.0f;
    __syncthreads();
    const float *roi = rois + (int64_t)r * 5;
    const int b = (int)roi[0];
    if (threadIdx.x == 0) axis_weights(roi[2] * sc, roi[4] * sc, H, wy_s, &rng_s[0], &rng_s[1]);
    if (threadIdx.x == 64) axis_weights(roi[1] * sc, roi[3] * sc, W, wx_s, &rng_s[2], &rng_s[3]);
    __syncthreads();
    const int h0 = rng_s[0], h1 = rng_s[1], w0 = rng_s[2], w1 = rng_s[3];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float norm = 1.0f / (float)(kRoiSamples * kRoiSamples);     // /4 per bin, /14, /14
    const bool empty = h1 < h0 || w1 < w0 || b < 0 || b >= B;
    if (!BWD) {
        // Forward: the weight of cell (h, w) of the patch is wy[h] * wx[w] for EVERY channel, so the patch is flattened
        // once into (weight, offset) tables in LDS and each wave then runs 4 channels at a time over it: lane e takes
        // cells e, e + 64, ...  -- 4 independent loads per step and no dependence between steps, where a row-by-row loop
        // per channel kept a single load in flight (measured 280 us for ONE large roi; this form is latency-hidden).
        // Patches above kPatchMax cells go in row tiles, partial sums accumulate in `out`.
        const T *f = (const T *)lv.feat[l] + (int64_t)(empty ? 0 : b) * C * H * W;
        const int64_t HWl = (int64_t)H * W;
        float *orow = out_or_dout + (int64_t)r * 4 * C + (int64_t)l * C;
        if (empty) {
            for (int c = threadIdx.x; c < C; c += 256) orow[c] = 0.0f;
            return;
        }
        const int ph = h1 - h0 + 1, pw = w1 - w0 + 1;
        int rpt = kPatchMax / pw;
        rpt = rpt < 1 ? 1 : (rpt > ph ? ph : rpt);
        for (int hb = h0; hb <= h1; hb += rpt) {
            const int rows = min(rpt, h1 - hb + 1), ne = rows * pw;
            __syncthreads();
            for (int e = threadIdx.x; e < ne; e += 256) {
                const int rr = e / pw, ww = e - rr * pw;
                wgt_s[e] = wy_s[hb + rr] * wx_s[w0 + ww];
                off_s[e] = (hb + rr) * W + w0 + ww;
            }
            __syncthreads();
            for (int c0 = wave * kRoiCG; c0 < C; c0 += 4 * kRoiCG) {
                float acc[kRoiCG];
                const T *fc[kRoiCG];
#pragma unroll
                for (int q = 0; q < kRoiCG; ++q) {
                    acc[q] = 0.0f;
                    fc[q] = f + (int64_t)(c0 + q < C ? c0 + q : C - 1) * HWl;
                }
#pragma unroll 4
                for (int e = lane; e < ne; e += 64) {
                    const float wg = wgt_s[e];
                    const int of = off_s[e];
#pragma unroll
                    for (int q = 0; q < kRoiCG; ++q) acc[q] = __builtin_fmaf(wg, to_f32<T>(fc[q][of]), acc[q]);
                }
                wave_sum_rows<kRoiCG>(acc);
                if (lane == 0) {
#pragma unroll
                    for (int q = 0; q < kRoiCG; ++q)
                        if (c0 + q < C) orow[c0 + q] = (hb == h0 ? 0.0f : orow[c0 + q]) + acc[q] * norm;
                }
            }
        }
    } else {
        if (empty) return;
        float *df = lv.dfeat[l] + (int64_t)b * C * H * W;
        for (int c = wave; c < C; c += 4) {
            const float g = out_or_dout[(int64_t)r * 4 * C + (int64_t)l * C + c] * norm;
            float *dc = df + (int64_t)c * H * W;
            for (int wq = w0 + lane; wq <= w1; wq += 64) {
                const float gx = g * wx_s[wq];
                for (int h = h0; h <= h1; ++h) {
                    const float v = gx * wy_s[h];
                    if (v != 0.0f) atomicAdd(&dc[(int64_t)h * W + wq], v);
                }
            }
        }
    }
}


// ---- channels-last (NHWC) features: what the inference encoder produces (encoder.FastEncoder keeps activations
// [B,H,W,C] bf16 end to end).  A feature cell is C contiguous channels, so a lane takes 16 bytes (8 bf16 / 4 fp32
// channels) of one cell, LPC = C / VEC lanes cover a cell and a wave takes 64 / LPC cells per load instruction -- every
// load is a full 16-byte lane access of contiguous memory, where the NCHW form reads 2-byte elements H*W apart.  Each
// lane accumulates its channels over its share of the roi's patch (weight wy[h] * wx[w], the same separable weights
// as above); the partial sums of the cell groups of a wave and of the four waves are then added in a fixed order.
// grid = (R, 4); block = 512.  Requires C % VEC == 0 and C / VEC a power of two <= 64 (or a multiple of 64).
template <typename T> struct NhwcVec;
template <> struct NhwcVec<float> {
    static constexpr int kVec = 4;
    static __device__ __forceinline__ void load(const float *p, float (&v)[4]) {
        const float4a t = *reinterpret_cast<const float4a *>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
};
template <> struct NhwcVec<bf16_t> {
    static constexpr int kVec = 8;
    static __device__ __forceinline__ void load(const bf16_t *p, float (&v)[8]) {
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        const u4 t = *reinterpret_cast<const u4 *>(p);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[2 * k] = __uint_as_float(t[k] << 16);
            v[2 * k + 1] = __uint_as_float(t[k] & 0xFFFF0000u);
        }
    }
};
template <> struct NhwcVec<f16_t> {
    static constexpr int kVec = 8;
    static __device__ __forceinline__ void load(const f16_t *p, float (&v)[8]) {
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        const h8 t = *reinterpret_cast<const h8 *>(p);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (float)t[k];
    }
};

constexpr int kNhwcWaves = 8;        // waves per (roi, level) workgroup
constexpr int kNhwcUnroll = 4;       // cells in flight per lane

template <typename T>
__global__ __launch_bounds__(64 * kNhwcWaves) void roialign4_mean_nhwc_kernel(RoiLevels lv, int B, int C,
                                                                              const float *__restrict__ rois, int R,
                                                                              float *__restrict__ out) {
    constexpr int VEC = NhwcVec<T>::kVec;
    constexpr int NT = 64 * kNhwcWaves;
    __shared__ float wy_s[kRoiMaxDim], wx_s[kRoiMaxDim];
    __shared__ float red_s[kNhwcWaves][64 * VEC];
    __shared__ int rng_s[4];
    const int r = blockIdx.x, l = blockIdx.y;
    const int H = lv.H[l], W = lv.W[l];
    const float sc = lv.scale[l];
    for (int i = threadIdx.x; i < H; i += NT) wy_s[i] = 0.0f;
    for (int i = threadIdx.x; i < W; i += NT) wx_s[i] = 0.0f;
    __syncthreads();
    const float *roi = rois + (int64_t)r * 5;
    const int b = (int)roi[0];
    if (threadIdx.x == 0) axis_weights(roi[2] * sc, roi[4] * sc, H, wy_s, &rng_s[0], &rng_s[1]);
    if (threadIdx.x == 64) axis_weights(roi[1] * sc, roi[3] * sc, W, wx_s, &rng_s[2], &rng_s[3]);
    __syncthreads();
    const int h0 = rng_s[0], h1 = rng_s[1], w0 = rng_s[2], w1 = rng_s[3];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float norm = 1.0f / (float)(kRoiSamples * kRoiSamples);
    float *orow = out + (int64_t)r * 4 * C + (int64_t)l * C;
    if (h1 < h0 || w1 < w0 || b < 0 || b >= B) {
        for (int c = threadIdx.x; c < C; c += NT) orow[c] = 0.0f;
        return;
    }
    const T *f = (const T *)lv.feat[l] + (int64_t)b * H * W * C;
    const int ph = h1 - h0 + 1, pw = w1 - w0 + 1, ncell = ph * pw;
    const int cblk = C < 64 * VEC ? C : 64 * VEC;              // channels one pass of the wave covers
    const int LPC = cblk / VEC, CPW = 64 / LPC;                // lanes per cell, cells per wave and load
    const int cs = lane / LPC, co = (lane - cs * LPC) * VEC;
    const int stride = kNhwcWaves * CPW;
    for (int c0 = 0; c0 < C; c0 += cblk) {
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.0f;
        // kNhwcUnroll independent cells in flight per lane: the loop is bound by load latency (a roi's patch is a few
        // hundred to a few thousand cells), not by bandwidth; cells past the patch are clamped and weighted 0
        for (int e = wave * CPW + cs; e < ncell; e += kNhwcUnroll * stride) {
            float v[kNhwcUnroll][VEC], g[kNhwcUnroll];
#pragma unroll
            for (int u = 0; u < kNhwcUnroll; ++u) {
                const int eu = e + u * stride;
                const int ec = eu < ncell ? eu : ncell - 1;
                const int rr = ec / pw, cc = ec - rr * pw;
                NhwcVec<T>::load(f + ((int64_t)(h0 + rr) * W + (w0 + cc)) * C + c0 + co, v[u]);
                g[u] = eu < ncell ? wy_s[h0 + rr] * wx_s[w0 + cc] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < kNhwcUnroll; ++u)
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] = __builtin_fmaf(g[u], v[u][k], acc[k]);
        }
        // fold the cell groups of the wave (lanes LPC apart hold the same channels), then the waves
        for (int d = LPC; d < 64; d <<= 1) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] += __shfl_xor(acc[k], d);
        }
        __syncthreads();
        if (cs == 0) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) red_s[wave][co + k] = acc[k];
        }
        __syncthreads();
        for (int c = threadIdx.x; c < cblk; c += NT) {
            float t = red_s[0][c];
#pragma unroll
            for (int w = 1; w < kNhwcWaves; ++w) t += red_s[w][c];
            orow[c0 + c] = t * norm;
        }
    }
}

}  // namespace dmm

static int roi_check(const void *const feat[4], const int H[4], const int W[4], const float scale[4], int B, int C,
                     const float *rois, int R) {
    if (B < 0 || C < 0 || R < 0) return DMM_ERR_BAD_ARG;
    if (R == 0 || C == 0) return DMM_OK;
    if (!feat || !H || !W || !scale || !rois) return DMM_ERR_BAD_ARG;
    for (int l = 0; l < 4; ++l) {
        if (!feat[l] || H[l] <= 0 || W[l] <= 0) return DMM_ERR_BAD_ARG;
        if (H[l] > dmm::kRoiMaxDim || W[l] > dmm::kRoiMaxDim) return DMM_ERR_UNSUPPORTED;
    }
    return -1;
}

extern "C" int dmm_roialign4_mean_fwd(const void *const feat[4], int dtype, int B, int C, const int H[4], const int W[4],
                                      const float scale[4], const float *rois, int R, float *out,
                                      dmm_stream_t stream) {
    const int rc = roi_check(feat, H, W, scale, B, C, rois, R);
    if (rc >= 0) return rc;
    if (!out) return DMM_ERR_BAD_ARG;
    dmm::RoiLevels lv;
    for (int l = 0; l < 4; ++l) {
        lv.feat[l] = feat[l]; lv.dfeat[l] = nullptr; lv.H[l] = H[l]; lv.W[l] = W[l]; lv.scale[l] = scale[l];
    }
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case DMM_F32:
            hipLaunchKernelGGL((dmm::roialign4_mean_kernel<float, false>), dim3(R, 4), dim3(256), 0, s, lv, B, C, rois, R,
                               out);
            break;
        case DMM_F16:
            hipLaunchKernelGGL((dmm::roialign4_mean_kernel<dmm::f16_t, false>), dim3(R, 4), dim3(256), 0, s, lv, B, C,
                               rois, R, out);
            break;
        case DMM_BF16:
            hipLaunchKernelGGL((dmm::roialign4_mean_kernel<dmm::bf16_t, false>), dim3(R, 4), dim3(256), 0, s, lv, B, C,
                               rois, R, out);
            break;
        default:
            return DMM_ERR_BAD_ARG;
    }
    return dmm::check_launch();
}

extern "C" int dmm_roialign4_mean_bwd(const float *dout, int B, int C, const int H[4], const int W[4],
                                      const float scale[4], const float *rois, int R, float *const dfeat[4],
                                      dmm_stream_t stream) {
    const int rc = roi_check((const void *const *)dfeat, H, W, scale, B, C, rois, R);
    if (rc >= 0) return rc;
    if (!dout) return DMM_ERR_BAD_ARG;
    dmm::RoiLevels lv;
    for (int l = 0; l < 4; ++l) {
        lv.feat[l] = nullptr; lv.dfeat[l] = dfeat[l]; lv.H[l] = H[l]; lv.W[l] = W[l]; lv.scale[l] = scale[l];
    }
    hipLaunchKernelGGL((dmm::roialign4_mean_kernel<float, true>), dim3(R, 4), dim3(256), 0, (hipStream_t)stream, lv, B, C,
                       rois, R, const_cast<float *>(dout));
    return dmm::check_launch();
}

// Channels-last (NHWC) form of dmm_roialign4_mean_fwd: feat[l] is [B, H[l], W[l], C] contiguous (a torch channels_last
// [B,C,H,W] tensor).  C % (16 / element size) == 0 and C / that a power of two (or a multiple of 64 x that).
extern "C" int dmm_roialign4_mean_nhwc_fwd(const void *const feat[4], int dtype, int B, int C, const int H[4],
                                           const int W[4], const float scale[4], const float *rois, int R, float *out,
                                           dmm_stream_t stream) {
    const int rc = roi_check(feat, H, W, scale, B, C, rois, R);
    if (rc >= 0) return rc;
    if (!out) return DMM_ERR_BAD_ARG;
    const int vec = dtype == DMM_F32 ? 4 : 8;
    if (C % vec != 0) return DMM_ERR_UNSUPPORTED;
    const int lpc = C / vec;
    if (!(lpc <= 64 ? (lpc & (lpc - 1)) == 0 : lpc % 64 == 0)) return DMM_ERR_UNSUPPORTED;
    for (int l = 0; l < 4; ++l)
        if (((uintptr_t)feat[l] & 15) != 0) return DMM_ERR_UNSUPPORTED;
    dmm::RoiLevels lv;
    for (int l = 0; l < 4; ++l) {
        lv.feat[l] = feat[l]; lv.dfeat[l] = nullptr; lv.H[l] = H[l]; lv.W[l] = W[l]; lv.scale[l] = scale[l];
    }
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case DMM_F32:
            hipLaunchKernelGGL((dmm::roialign4_mean_nhwc_kernel<float>), dim3(R, 4), dim3(64 * dmm::kNhwcWaves), 0, s, lv, B, C, rois, R, out);
            break;
        case DMM_F16:
            hipLaunchKernelGGL((dmm::roialign4_mean_nhwc_kernel<dmm::f16_t>), dim3(R, 4), dim3(64 * dmm::kNhwcWaves), 0, s, lv, B, C, rois,
                               R, out);
            break;
        case DMM_BF16:
            hipLaunchKernelGGL((dmm::roialign4_mean_nhwc_kernel<dmm::bf16_t>), dim3(R, 4), dim3(64 * dmm::kNhwcWaves), 0, s, lv, B, C, rois,
                               R, out);
            break;
        default:
            return DMM_ERR_BAD_ARG;
    }
    return dmm::check_launch();
}
