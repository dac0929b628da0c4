// dmm_encoder_ops.hip -- the elementwise epilogue of the inference encoder on gfx950 (channels-last bf16).
//
// Reference: the conv -> BatchNorm -> ReLU stacks of the encoder, dmm/modules/base.py:43-54 (prop heads), :35-42
// (skip projections + bn, model_encoder.py:137-140) and the torchvision Bottleneck / BasicBlock tails reached through
// dmm/modules/vision.py:6-38 (out = relu(bn3(conv3(x)) + identity)).  With BatchNorm folded into the convolution
// (encoder.fold_batchnorm) what is left after each MIOpen / hipBLASLt contraction is
//     y = act(x + bias[c] (+ residual))
// which eager PyTorch runs as 2-3 separate launches (bias add, residual add, clamp): 134 of the 349 launches and 29 %
// of the device time of the ResNet-50 forward at 8 x 255 x 255 (profiles/r02_encoder_kernel_table_nchw_eager.md).
// Here it is ONE in-place pass: 16-byte lane loads (8 bf16), fp32 arithmetic, one rounding.
//
// Roofline: HBM (2-3 x rows*C*2 bytes); the tensors are small (<= 16 MB), so it is launch / L2 bound in practice.
#include "dmm_common.h"

namespace dmm {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t bf16_rne(float v) {
    const uint32_t u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// x: [rows, C] bf16 (channels-last activation viewed 2-D), C % 8 == 0.  One thread = 8 consecutive channels.
template <bool RES, bool RELU>
__global__ __launch_bounds__(256) void bias_act_bf16_kernel(uint16_t *__restrict__ x, const float *__restrict__ bias,
                                                            const uint16_t *__restrict__ res, int64_t n8, int c8) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const int c = (int)(i % c8) * 8;
    u32x4 v = reinterpret_cast<const u32x4 *>(x)[i];
    u32x4 r;
    if (RES) r = reinterpret_cast<const u32x4 *>(res)[i];
    const float4 b0 = bias ? *reinterpret_cast<const float4 *>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 b1 = bias ? *reinterpret_cast<const float4 *>(bias + c + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float lo = __uint_as_float(v[k] << 16) + bb[2 * k];
        float hi = __uint_as_float(v[k] & 0xFFFF0000u) + bb[2 * k + 1];
        if (RES) {
            lo = lo + __uint_as_float(r[k] << 16);
            hi = hi + __uint_as_float(r[k] & 0xFFFF0000u);
        }
        if (RELU) {
            lo = lo > 0.0f ? lo : 0.0f;
            hi = hi > 0.0f ? hi : 0.0f;
        }
        o[k] = bf16_rne(lo) | (bf16_rne(hi) << 16);
    }
    reinterpret_cast<u32x4 *>(x)[i] = o;
}

// C not a multiple of 8 (tiny heads, e.g. hidden_size 8 -> 2-channel skips): one element per thread.
template <bool RES, bool RELU>
__global__ __launch_bounds__(256) void bias_act_bf16_scalar_kernel(uint16_t *__restrict__ x, const float *__restrict__ bias,
                                                                   const uint16_t *__restrict__ res, int64_t n, int C) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = __uint_as_float((uint32_t)x[i] << 16) + (bias ? bias[i % C] : 0.0f);
    if (RES) v = v + __uint_as_float((uint32_t)res[i] << 16);
    if (RELU) v = v > 0.0f ? v : 0.0f;
    x[i] = (uint16_t)bf16_rne(v);
}

// Patch matrix of a 3x3 / pad 1 convolution on a channels-last activation: cols[(b, ho, wo), (kh, kw, c)] =
// x[b, s*ho + kh - 1, s*wo + kw - 1, c] (zero outside the image).  One thread = 8 consecutive channels of one tap (16-byte
// load, 16-byte store); the nine taps of an output pixel are neighbouring threads, so the input stays in L2 (each element
// is read 9 / s^2 times) and the stores of a row of the matrix are contiguous.
__global__ __launch_bounds__(256) void im2col3x3_bf16_kernel(const uint16_t *__restrict__ x, uint16_t *__restrict__ cols,
                                                             int H, int W, int c8, int Ho, int Wo, int stride, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;           // over rows * 9 * c8
    if (i >= n) return;
    const int c = (int)(i % c8);
    int64_t t = i / c8;
    const int tap = (int)(t % 9);
    t /= 9;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const int64_t b = t / Ho;
    const int hi = ho * stride + tap / 3 - 1, wi = wo * stride + tap % 3 - 1;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (hi >= 0 && hi < H && wi >= 0 && wi < W) v = reinterpret_cast<const u32x4 *>(x)[((b * H + hi) * W + wi) * c8 + c];
    reinterpret_cast<u32x4 *>(cols)[i] = v;
}

// Stem tail: relu(x + bias) followed by the 3x3 / stride 2 / padding 1 max-pool, in one pass over the convolution's output.
// x + bias, relu and the bf16 rounding are non-decreasing, so they commute with the maximum: the window maximum of the RAW
// values is taken first (exact in bf16), then bias, relu and ONE rounding -- bit identical to rounding every element
// first and pooling after, without writing and re-reading the 4x larger pre-pool activation.
__global__ __launch_bounds__(256) void bias_relu_maxpool_bf16_kernel(const uint16_t *__restrict__ x,
                                                                     const float *__restrict__ bias,
                                                                     uint16_t *__restrict__ y, int H, int W, int c8, int Ho,
                                                                     int Wo, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;           // over B * Ho * Wo * c8
    if (i >= n) return;
    const int c = (int)(i % c8);
    int64_t t = i / c8;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const int64_t b = t / Ho;
    float m[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k] = -__builtin_inff();
#pragma unroll
    for (int dh = -1; dh <= 1; ++dh) {
        const int hi = 2 * ho + dh;
        if (hi < 0 || hi >= H) continue;
#pragma unroll
        for (int dw = -1; dw <= 1; ++dw) {
            const int wi = 2 * wo + dw;
            if (wi < 0 || wi >= W) continue;
            const u32x4 v = reinterpret_cast<const u32x4 *>(x)[((b * H + hi) * W + wi) * c8 + c];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float lo = __uint_as_float(v[k] << 16), hi_ = __uint_as_float(v[k] & 0xFFFF0000u);
                m[2 * k] = lo > m[2 * k] ? lo : m[2 * k];
                m[2 * k + 1] = hi_ > m[2 * k + 1] ? hi_ : m[2 * k + 1];
            }
        }
    }
    const float4 b0 = *reinterpret_cast<const float4 *>(bias + 8 * c), b1 = *reinterpret_cast<const float4 *>(bias + 8 * c + 4);
    const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float lo = m[2 * k] + bb[2 * k], hi_ = m[2 * k + 1] + bb[2 * k + 1];
        lo = lo > 0.0f ? lo : 0.0f;
        hi_ = hi_ > 0.0f ? hi_ : 0.0f;
        o[k] = bf16_rne(lo) | (bf16_rne(hi_) << 16);
    }
    reinterpret_cast<u32x4 *>(y)[i] = o;
}

}  // namespace dmm

extern "C" int dmm_bias_relu_maxpool_bf16(const void *x, const float *bias, int B, int H, int W, int C, void *y,
                                          dmm_stream_t stream) {
    if (B < 0 || H <= 0 || W <= 0 || C <= 0) return DMM_ERR_BAD_ARG;
    if (B == 0) return DMM_OK;
    if (!x || !bias || !y) return DMM_ERR_BAD_ARG;
    if (C & 7) return DMM_ERR_UNSUPPORTED;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int64_t n = (int64_t)B * Ho * Wo * (C / 8), blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffLL) return DMM_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(dmm::bias_relu_maxpool_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)x, bias, (uint16_t *)y, H, W, C / 8, Ho, Wo, n);
    return dmm::check_launch();
}

extern "C" int dmm_im2col3x3_bf16(const void *x, int B, int H, int W, int C, int stride, void *cols, dmm_stream_t stream) {
    if (B < 0 || H <= 0 || W <= 0 || C <= 0 || (stride != 1 && stride != 2)) return DMM_ERR_BAD_ARG;
    if (B == 0) return DMM_OK;
    if (!x || !cols) return DMM_ERR_BAD_ARG;
    if (C & 7) return DMM_ERR_UNSUPPORTED;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;     // (H + 2 - 3) / s + 1
    const int64_t n = (int64_t)B * Ho * Wo * 9 * (C / 8), blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffLL) return DMM_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(dmm::im2col3x3_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)x, (uint16_t *)cols, H, W, C / 8, Ho, Wo, stride, n);
    return dmm::check_launch();
}

extern "C" int dmm_bias_act_bf16(void *x, const float *bias, const void *residual, int64_t rows, int C, int relu,
                                 dmm_stream_t stream) {
    if (rows < 0 || C <= 0) return DMM_ERR_BAD_ARG;
    if (rows == 0) return DMM_OK;
    if (!x) return DMM_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    uint16_t *xp = (uint16_t *)x;
    const uint16_t *rp = (const uint16_t *)residual;
    if (C & 7) {
        const int64_t n = rows * (int64_t)C, nb = (n + 255) / 256;
        if (nb > 0x7fffffffLL) return DMM_ERR_UNSUPPORTED;
#define DMM_BAS(RES_, RELU_)                                                                                         \
    hipLaunchKernelGGL((dmm::bias_act_bf16_scalar_kernel<RES_, RELU_>), dim3((unsigned)nb), dim3(256), 0, s, xp, bias, \
                       rp, n, C)
        if (residual) { if (relu) DMM_BAS(true, true); else DMM_BAS(true, false); }
        else { if (relu) DMM_BAS(false, true); else DMM_BAS(false, false); }
#undef DMM_BAS
        return dmm::check_launch();
    }
    const int64_t n8 = rows * (int64_t)(C / 8);
    const int64_t blocks = (n8 + 255) / 256;
    if (blocks > 0x7fffffffLL) return DMM_ERR_UNSUPPORTED;
#define DMM_BA(RES_, RELU_)                                                                                          \
    hipLaunchKernelGGL((dmm::bias_act_bf16_kernel<RES_, RELU_>), dim3((unsigned)blocks), dim3(256), 0, s, xp, bias, rp, \
                       n8, C / 8)
    if (residual) { if (relu) DMM_BA(true, true); else DMM_BA(true, false); }
    else { if (relu) DMM_BA(false, true); else DMM_BA(false, false); }
#undef DMM_BA
    return dmm::check_launch();
}
