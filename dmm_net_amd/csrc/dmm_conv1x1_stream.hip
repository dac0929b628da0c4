// dmm_conv1x1_stream.hip -- the HBM-bound 1x1 convolutions of the channels-last inference encoder as a streaming MFMA kernel.
//
//   y[rows, cout] = relu?( x[rows, cin] . w[cin, cout] + bias[cout] (+ residual[rows, cout]) )        (dmm_conv1x1_bf16)
//
// At the stride-4 level of the ResNet bodies (dmm/modules/vision.py:6-38: 64 -> 64 and 256 -> 64 on the [B, H/4, W/4] map)
// the product is a few GFLOP on tens of MB of activations: 0.5-1.5 us of MFMA time against 10-20 us of HBM time.  The library GEMM (hipBLASLt) reaches 1.5-3.8 TB/s
// there (profiles/r04_encoder_layer_table.md: 64 -> 64 1.45 TB/s, 256 -> 64 2.2 TB/s at 16 images of 255 x 448): its tiles
// are shaped for compute-bound problems.  This kernel is shaped for the stream:
//   * the WEIGHTS live in registers for the life of a wave (cin x cout <= 32 K elements: <= 128 VGPRs per lane as MFMA
//     fragments; staged once per workgroup through LDS), so the inner loop touches memory only for activations;
//   * the product is computed TRANSPOSED (channels = MFMA rows, pixels = MFMA columns): a lane then owns one pixel and
//     runs of 4 consecutive output channels (C/D layout of v_mfma_f32_32x32x16_bf16: row = (reg & 3) + 8 (reg >> 2) +
//     4 (lane >> 5), col = lane & 31), i.e. 8-byte pieces of the pixel's channels-last output row -- stored straight from
//     the accumulators, the residual read the same way, no LDS turn for the epilogue (fine while a pixel's slice is <= 128
//     bytes; for wide outputs the pieces lie too far apart -- see conv1x1_stream_launch);
//   * the activation operand is ONE 16-byte load per lane and k-step straight from global memory (lane = pixel, 8
//     consecutive input channels): rows of x are consecutive, a 32-pixel tile is one contiguous 32 * cin * 2 bytes;
//   * waves are persistent (grid = one resident round of workgroups, tiles strided) so the weight fragments are built once;
//     256-channel inputs leave one wave per SIMD, so there the next tile's loads are issued before the current tile's
//     arithmetic and stores (and the first tile's before the weights are staged).
// fp32 accumulation, bias added in fp32 (it is the accumulator's initial value), one rounding to bfloat16 -- the contract of
// dmm_conv1x1_bf16; the summation order differs from the library's, so results agree to fp32 rounding, not bit for bit.
#include "dmm_common.h"

namespace dmm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kStreamThreads = 256;

template <int CIN, int COUT, int SPLIT, bool PREFETCH>
__global__ __launch_bounds__(kStreamThreads) void conv1x1_stream_kernel(
    const __bf16 *__restrict__ x, const __bf16 *__restrict__ w, const float *__restrict__ bias,
    const __bf16 *__restrict__ residual, int64_t rows, int relu, __bf16 *__restrict__ y, int tiles) {
    constexpr int KS = CIN / 16;                                     // k-steps of the 32x32x16 MFMA
    constexpr int CW = COUT / SPLIT;                                 // output channels of one wave
    constexpr int NB = CW / 32;                                      // 32-channel blocks of one wave
    static_assert(CIN % 16 == 0 && CW % 32 == 0 && (kStreamThreads / 64) % SPLIT == 0, "shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char stream_lds[];
    __bf16 *wl = reinterpret_cast<__bf16 *>(stream_lds);             // [CIN][COUT]
    float *bl = reinterpret_cast<float *>(stream_lds + sizeof(__bf16) * CIN * COUT);   // [COUT]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 31, h = lane >> 5;
    constexpr int TW = (kStreamThreads / 64) / SPLIT;                // pixel tiles a workgroup has in flight
    const int tstep = gridDim.x * TW;
    auto load_x = [&](int t, bf16x8 (&f)[KS]) {                      // tail tile: re-read the last row, store nothing
        const int64_t p = (int64_t)t * 32 + m;
        const __bf16 *xr = x + (p < rows ? p : rows - 1) * CIN + 8 * h;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) f[kk] = *reinterpret_cast<const bf16x8 *>(xr + 16 * kk);
    };
    bf16x8 xa[KS];
    int t = blockIdx.x * TW + wave / SPLIT;
    if (PREFETCH && t < tiles) load_x(t, xa);                        // in flight while the weights are staged
    for (int i = threadIdx.x * 8; i < CIN * COUT; i += kStreamThreads * 8)
        *reinterpret_cast<uint4 *>(wl + i) = *reinterpret_cast<const uint4 *>(w + i);
    for (int i = threadIdx.x; i < COUT; i += kStreamThreads) bl[i] = bias[i];
    __syncthreads();
    const int c0 = (wave % SPLIT) * CW;
    // weight fragments (the MFMA's A operand: row = channel c0 + 32 nb + m, k = 16 kk + 8 h + j)
    bf16x8 wf[NB][KS];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
#pragma unroll
            for (int j = 0; j < 8; ++j) wf[nb][kk][j] = wl[(16 * kk + 8 * h + j) * COUT + c0 + 32 * nb + m];
    // one tile; with PREFETCH the NEXT tile's activations are requested before this tile's arithmetic and stores (the
    // 256-channel inputs leave one wave per SIMD: nothing else would be loading meanwhile)
    auto tile = [&](int t, bf16x8 (&xf)[KS], bf16x8 (&xn)[KS]) {
        const int64_t p = (int64_t)t * 32 + m;
        const int64_t pl = p < rows ? p : rows - 1;
        if (PREFETCH && t + tstep < tiles) load_x(t + tstep, xn);
        bf16x4 rs[NB][4];
        if (residual) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    rs[nb][g] = *reinterpret_cast<const bf16x4 *>(residual + pl * COUT + c0 + 32 * nb + 8 * g + 4 * h);
        }
        f32x16 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b4 = *reinterpret_cast<const float4 *>(bl + c0 + 32 * nb + 8 * g + 4 * h);
                acc[nb][4 * g] = b4.x; acc[nb][4 * g + 1] = b4.y; acc[nb][4 * g + 2] = b4.z; acc[nb][4 * g + 3] = b4.w;
            }
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nb][kk], xf[kk], acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4] = {acc[nb][4 * g], acc[nb][4 * g + 1], acc[nb][4 * g + 2], acc[nb][4 * g + 3]};
                if (residual) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = v[i] + (float)rs[nb][g][i];
                }
                if (relu) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.0f ? v[i] : 0.0f;
                }
                bf16x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (__bf16)v[i];
                if (p < rows) *reinterpret_cast<bf16x4 *>(y + p * COUT + c0 + 32 * nb + 8 * g + 4 * h) = o;
            }
    };
    if constexpr (PREFETCH) {
        bf16x8 xb[KS];
        for (; t < tiles; t += 2 * tstep) {
            tile(t, xa, xb);
            if (t + tstep < tiles) tile(t + tstep, xb, xa);
        }
    } else {
        for (; t < tiles; t += tstep) {
            load_x(t, xa);
            tile(t, xa, xa);
        }
    }
}

template <int CIN, int COUT, int SPLIT, int WGS_PER_CU, bool PREFETCH = (CIN >= 256)>
static int launch_stream(const void *x, const void *w, const float *bias, const void *residual, int64_t rows, int relu,
                         void *y, hipStream_t stream) {
    constexpr int TW = (kStreamThreads / 64) / SPLIT;
    const int64_t tiles64 = (rows + 31) / 32;
    if (tiles64 > 0x7fffffff) return DMM_ERR_UNSUPPORTED;
    const int tiles = (int)tiles64;
    const size_t lds = sizeof(__bf16) * CIN * COUT + sizeof(float) * COUT;
    // persistent waves: one resident round of workgroups (the occupancy the register count really allows)
    static int per_cu = 0;
    if (per_cu == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)conv1x1_stream_kernel<CIN, COUT, SPLIT, PREFETCH>,
                                                         kStreamThreads, lds) != hipSuccess || n < 1) {
            (void)hipGetLastError();
            n = 1;
        }
        per_cu = n > WGS_PER_CU ? WGS_PER_CU : n;
    }
    // ~4 tiles per wave (the weight staging amortised) but never fewer workgroups than two resident rounds when the
    // problem has them: under concurrent streams (the frame loop runs the encoder beside the frame steps and the heads
    // beside the body) a grid of exactly one resident round left late-starting workgroups with a full share of tiles
    int grid = (tiles + TW - 1) / TW;
    const int want = (grid + 3) / 4;
    const int floor_wgs = 2 * 256 * per_cu;
    grid = want > floor_wgs ? want : (grid < floor_wgs ? grid : floor_wgs);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)conv1x1_stream_kernel<CIN, COUT, SPLIT, PREFETCH>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_last_hip_error((int)e); return DMM_ERR_LAUNCH; }
    }
    hipLaunchKernelGGL((conv1x1_stream_kernel<CIN, COUT, SPLIT, PREFETCH>), dim3(grid), dim3(kStreamThreads), lds, stream,
                       (const __bf16 *)x, (const __bf16 *)w, bias, (const __bf16 *)residual, rows, relu, (__bf16 *)y, tiles);
    return check_launch();
}

// DMM_ERR_UNSUPPORTED (nothing launched) unless DMM_OPT_CONV1X1_STREAM = 1 and the shape is one it is built for.
// Built for the REDUCING / narrow products of the stride-4 level, where the stream is mostly reads; alone on the chip:
//   16 / 48 images of 255 x 448 (114 688 / 344 064 rows):  64 -> 64   21.0 / 58.6 us  ->  10.1 / 22.2 us
//                                                          256 -> 64   33.3 / 93.3 us  ->  20.8 / 44.2 us  (5.0 TB/s)
// i.e. -24 us of 1684 (16 images) and -140 us of 3850 (48 images) per ResNet-50 forward.  OFF by default all the same: the
// evaluator's frame loop, where the encoder's graphs run beside the frame steps, did not move (0.406-0.412 ms per step
// without, 0.410-0.421 with, same box) -- its pace is not set by these launches -- and the library GEMM is the path with
// the longer record.
// Measured and NOT taken: 64 -> 256 and 256 -> 128 (output-heavy; 8-byte pieces 512 bytes apart, or turned through LDS into
// 16-byte lane stores: 23.5 / 55.0 us and 31.1 / 74.7 us against the library's 20.2 / 51.5 and 23.5 / 82.1).
int conv1x1_stream_launch(const void *x, const void *w, const float *bias, const void *residual, int64_t rows, int cin,
                          int cout, int relu, void *y, hipStream_t stream) {
    if (opt(DMM_OPT_CONV1X1_STREAM) != 1) return DMM_ERR_UNSUPPORTED;
    if (cin == 64 && cout == 64 && rows >= 32768) return launch_stream<64, 64, 1, 4>(x, w, bias, residual, rows, relu, y, stream);
    if (cin == 256 && cout == 64 && rows >= 98304) return launch_stream<256, 64, 1, 2>(x, w, bias, residual, rows, relu, y, stream);
    return DMM_ERR_UNSUPPORTED;
}

}  // namespace dmm
