// dmm_cosine_lanes.hip -- the one-launch feature similarity with the D axis spread over the lanes of a wave.
//
// Same result as cosine_fused_kernel (dmm_cosine.hip), i.e. the reference's get_cosine_score
// (dmm/utils/match_helper.py:51-64) under torch 2.10 semantics, bit for bit:
//     cos[m, n] = sum_d RN( (t[m,d] / max(||t[m]||, 1e-8)) * (p[n,d] / max(||p[n]||, 1e-8)) )
// with the norm in the order of ATen's 2-norm fast path and the sum over d in the order ATen reduces the [D, N] slab
// of products (column classes of dmm_torch_order.h).
//
// Why another mapping.  With one thread per output the D-long chain of an output is 2 LDS reads per multiply-add
// (4 wave-wide 16-byte reads per output): at config 5 (N = 200, M = 20) the LDS pipe, not the VALU, set the pace, and
// the load / norm / divide / dot phases of a 140 KB tile ran one after the other with one block per CU (0.33 ms per
// 512 frames).  ATen's order has more parallelism than one chain per output: the 16-element blocks of the cascade
// (level_step 16) are independent sums, only their combination is sequential.  So here
//   * a lane owns ONE 16-element block of a proposal row (class A columns: d = 16k .. 16k+15; class B columns, whose
//     four ILP chains take every fourth d: d = 64(k/4) + 4i + k%4) of 8 columns at a time, already divided by the
//     norm, in registers -- D/16 lanes per column, 8 columns per step;
//   * the normalised templates sit in LDS once per block; per template a lane reads its 16 values ONCE for all its
//     columns (0.5 wave-wide reads per output), multiplies-adds them as packed pairs of columns, and leaves one
//     partial sum per output;
//   * the partial sums of 7-8 templates x 8 columns go through LDS to one lane per output, which adds them in the
//     cascade's order (32 adds for D = 512).
// Waves are autonomous after the template stage (no block barrier in the column loop), so the loads of one wave
// overlap the arithmetic of the others.  The division by the norm is the hardware's own refinement sequence with the
// denominator part hoisted out of the row (RowDiv below), falling back to a true division where scaling would matter.
#include "dmm_cosine_lanes.h"

namespace dmm {

// grid = (parts, B), block = 64 * nw threads; wave (part, wave) takes the steps part * nw + wave, + parts * nw, ...
// zero_ptr / zero_words: an unrelated buffer the launch clears on the side (dmm_match_forward: the IoU count tables the
// next kernel accumulates into -- saves the memset node in front of it)
template <int LPC>
__global__ __launch_bounds__(512) void cosine_lanes_kernel(const float *__restrict__ feat_t,
                                                           const float *__restrict__ feat_p, int N, int M,
                                                           float *__restrict__ cos_out, int32_t *__restrict__ zero_ptr,
                                                           int64_t zero_words, const int32_t *__restrict__ n_valid) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (zero_ptr) {
        const int64_t nblk = (int64_t)gridDim.x * gridDim.y, blk = blockIdx.x + (int64_t)gridDim.x * blockIdx.y;
        const int64_t per = (zero_words + nblk - 1) / nblk, lo = blk * per;
        const int64_t hi = lo + per < zero_words ? lo + per : zero_words;
        for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) zero_ptr[i] = 0;
    }
    cosine_lanes_body<LPC>(feat_t, feat_p, N, M, cos_out, lds, blockIdx.x, gridDim.x, blockIdx.y, blockDim.x >> 6, n_valid);
}

template <int LPC>
static int launch_lanes(const float *feat_t, const float *feat_p, int B, int N, int M, float *cos_out, int32_t *zero_ptr,
                        int64_t zero_words, const int32_t *n_valid, hipStream_t stream) {
    const LanesGeom g = lanes_geom<LPC>(B, N, M);
    if (!g.ok) return DMM_ERR_UNSUPPORTED;
    if (g.lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)cosine_lanes_kernel<LPC>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)g.lds);
        if (e != hipSuccess) { set_last_hip_error((int)e); return DMM_ERR_LAUNCH; }
    }
    hipLaunchKernelGGL((cosine_lanes_kernel<LPC>), dim3(g.parts, B), dim3(64 * g.nw), g.lds, stream, feat_t, feat_p, N, M,
                       cos_out, zero_ptr, zero_words, n_valid);
    return check_launch();
}

// DMM_ERR_UNSUPPORTED outside the envelope (D in {256, 512, 1024}); the caller then takes the tile kernel.
// n_valid (may be NULL): live proposals per frame -- every frame is reduced in the order of ITS count (see the body).
int cosine_lanes_launch(const float *feat_t, const float *feat_p, int B, int N, int M, int D, float *cos_out,
                        hipStream_t stream, int32_t *zero_ptr, int64_t zero_words, const int32_t *n_valid) {
    if (N < 2 || M < 1 || M > 32 || B > 65535) return DMM_ERR_UNSUPPORTED;
    switch (D) {
        case 256: return launch_lanes<16>(feat_t, feat_p, B, N, M, cos_out, zero_ptr, zero_words, n_valid, stream);
        case 512: return launch_lanes<32>(feat_t, feat_p, B, N, M, cos_out, zero_ptr, zero_words, n_valid, stream);
        case 1024: return launch_lanes<64>(feat_t, feat_p, B, N, M, cos_out, zero_ptr, zero_words, n_valid, stream);
        default: return DMM_ERR_UNSUPPORTED;
    }
}

}  // namespace dmm
