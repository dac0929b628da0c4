// dmm_pack.hip -- threshold-and-pack mask planes into the library's 1-bit "ballot layout" (DMM_PACKED1).
//
// The cost of the matching layer is streaming the mask planes (dmm_cost.hip); the only thing it does with a pixel is
// `x > 0.5` (reference dmm/utils/match_helper.py:16-17).  Planes that are matched more than once (training: templates
// AND targets; any re-use across calls) or that this library produces itself (dmm_paste_masks_f32) can be handed to
// dmm_iou_counts already packed: 32x fewer bytes for fp32 sources, identical integer tables.
// One wave packs 8 x 256 pixels per step: eight lane loads in flight, 32 v_cmp ballots, the 32 words go to LDS.
// Roofline: HBM (read side).
#include <stdlib.h>

#include "dmm_common.h"

namespace dmm {

// The U blocks (U x 256 pixels) a wave packs per iteration; the loads of the next iteration are issued before the ballots.
template <typename T, int U>
__device__ __forceinline__ void pack_load(const T *src, int HW, int q0, int lane, float (&v)[U][4]) {
    if ((q0 + U) * 256 <= HW) {                 // wave-uniform: the U loads issue back to back, no per-load branch
#pragma unroll
        for (int u = 0; u < U; ++u) MaskIO<T>::load4(src + (q0 + u) * 256 + lane * 4, v[u]);
        return;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int x = (q0 + u) * 256 + lane * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[u][k] = x + k < HW ? MaskIO<T>::load1(src + x + k) : 0.0f;
    }
}

constexpr int kPackSeg = 256;        // blocks of 256 pixels a workgroup packs per segment: 64 Ki pixels -> 8 KiB of words

// grid = (segments, planes); block = 256.  The words of a segment are STAGED IN LDS and written out as one contiguous
// 8 KiB burst at the end: a 128-byte store after every 4 KiB of reads (the first version) cost 20 % of the read
// stream -- the same kernel with the stores removed reads at 6.75 TB/s, with them at 5.4 -- although the stores are
// 3 % of the bytes.
template <typename T, int SEG = kPackSeg, int U = 4>
__global__ __launch_bounds__(256) void pack_masks_kernel(const T *__restrict__ masks, int HW, int64_t plane_stride,
                                                         unsigned long long *__restrict__ packed,
                                                         int64_t packed_stride) {
    constexpr int kPackSeg = SEG;
    __shared__ unsigned long long stage[kPackSeg * 4];
    const int64_t plane = blockIdx.y;
    const T *src = masks + plane * plane_stride;
    unsigned long long *dst = packed + plane * packed_stride;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nblocks = (HW + 255) / 256;
    const int seg0 = blockIdx.x * kPackSeg;
    const int seg_end = min(nblocks, seg0 + kPackSeg);
    int q0 = seg0 + wave * U;
    float cur[U][4], nxt[U][4];
    if (q0 < seg_end) pack_load<T, U>(src, HW, q0, lane, cur);
    for (; q0 < seg_end; q0 += 4 * U) {
        const bool more = q0 + 4 * U < seg_end;
        if (more) pack_load<T, U>(src, HW, q0 + 4 * U, lane, nxt);
        unsigned long long w = 0;
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned long long bal = __ballot(cur[u][k] > 0.5f);
                w = lane == 4 * u + k ? bal : w;
            }
        if (lane < 4 * U && 4 * (q0 - seg0) + lane < 4 * kPackSeg) stage[4 * (q0 - seg0) + lane] = w;   // (blocks past the plane are never copied out)
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) cur[u][k] = nxt[u][k];
        }
    }
    __syncthreads();
    const int nwords = 4 * (seg_end - seg0);
    for (int i = threadIdx.x; i < nwords; i += 256) dst[4 * seg0 + i] = stage[i];
}

template <typename T>
static int pack_typed(const T *masks, int64_t planes, int HW, int64_t plane_stride, unsigned long long *packed,
                      int64_t packed_stride, hipStream_t stream) {
    const int nblocks = (HW + 255) / 256;
    // segment length x blocks in flight per wave, measured at 51 200 planes of 255 x 255 fp32 (TB/s read): 256x4 5.11 (round
    // 2's form), 256x8 5.49, 64x4 5.40, 64x8 5.20, 128x8 5.54, 32x8 4.74 -- DMM_OPT_PACK_VARIANT = 0 is round 2's form
    const int variant = opt(DMM_OPT_PACK_VARIANT);
    for (int64_t p0 = 0; p0 < planes; p0 += 65535) {
        const int64_t np = planes - p0 < 65535 ? planes - p0 : 65535;
#define DMM_PACK_LAUNCH(SEG_, U_)                                                                                       \
    hipLaunchKernelGGL((pack_masks_kernel<T, SEG_, U_>), dim3((nblocks + SEG_ - 1) / SEG_, (unsigned)np), dim3(256), 0, \
                       stream, masks + p0 * plane_stride, HW, plane_stride, packed + p0 * packed_stride, packed_stride)
        switch (variant) {
            case 0: DMM_PACK_LAUNCH(256, 4); break;
            default: DMM_PACK_LAUNCH(128, 8); break;
        }
#undef DMM_PACK_LAUNCH
    }
    return check_launch();
}

}  // namespace dmm

extern "C" int64_t dmm_pack_words(int HW) { return HW < 0 ? 0 : 4 * ((int64_t)(HW + 255) / 256); }

extern "C" int dmm_pack_masks(const void *masks, int dtype, int64_t planes, int HW, int64_t plane_stride,
                              uint64_t *packed, int64_t packed_stride, dmm_stream_t stream) {
    if (planes < 0 || HW < 0) return DMM_ERR_BAD_ARG;
    if (planes == 0 || HW == 0) return DMM_OK;
    if (!masks || !packed || plane_stride < HW || packed_stride < dmm_pack_words(HW)) return DMM_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    unsigned long long *out = reinterpret_cast<unsigned long long *>(packed);
    switch (dtype) {
        case DMM_F32: return dmm::pack_typed<float>((const float *)masks, planes, HW, plane_stride, out, packed_stride, s);
        case DMM_F16:
            return dmm::pack_typed<dmm::f16_t>((const dmm::f16_t *)masks, planes, HW, plane_stride, out, packed_stride, s);
        case DMM_BF16:
            return dmm::pack_typed<dmm::bf16_t>((const dmm::bf16_t *)masks, planes, HW, plane_stride, out, packed_stride, s);
        default: return DMM_ERR_BAD_ARG;
    }
}
