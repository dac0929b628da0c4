// dmm_pack.hip -- threshold-and-pack mask planes into the library's 1-bit "ballot layout" (DMM_PACKED1).
//
// The cost of the matching layer is streaming the mask planes (dmm_cost.hip); the only thing it does with a pixel is
// `x > 0.5` (reference dmm/utils/match_helper.py:16-17).  Planes that are matched more than once (training: templates
// AND targets; any re-use across calls) or that this library produces itself (dmm_paste_masks_f32) can be handed to
// dmm_iou_counts already packed: 32x fewer bytes for fp32 sources, identical integer tables.
// One wave packs 256 pixels per step: a 16-byte lane load, four v_cmp ballots, lanes 0..3 store the four words.
// Roofline: HBM (read side).
#include "dmm_common.h"

namespace dmm {

// grid = (blocks of 256 pixels / 4 per workgroup ..., planes)
template <typename T>
__global__ __launch_bounds__(256) void pack_masks_kernel(const T *__restrict__ masks, int HW, int64_t plane_stride,
                                                         unsigned long long *__restrict__ packed,
                                                         int64_t packed_stride) {
    const int64_t plane = blockIdx.y;
    const T *src = masks + plane * plane_stride;
    unsigned long long *dst = packed + plane * packed_stride;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nblocks = (HW + 255) / 256;
    for (int q = blockIdx.x * 4 + wave; q < nblocks; q += gridDim.x * 4) {
        const int x = q * 256 + lane * 4;
        float v[4];
        if (x + 3 < HW) {
            MaskIO<T>::load4(src + x, v);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = x + k < HW ? MaskIO<T>::load1(src + x + k) : 0.0f;
        }
        const unsigned long long b0 = __ballot(v[0] > 0.5f), b1 = __ballot(v[1] > 0.5f);
        const unsigned long long b2 = __ballot(v[2] > 0.5f), b3 = __ballot(v[3] > 0.5f);
        if (lane < 4) dst[4 * q + lane] = lane == 0 ? b0 : (lane == 1 ? b1 : (lane == 2 ? b2 : b3));
    }
}

template <typename T>
static int pack_typed(const T *masks, int64_t planes, int HW, int64_t plane_stride, unsigned long long *packed,
                      int64_t packed_stride, hipStream_t stream) {
    const int nblocks = (HW + 255) / 256;
    int gx = (nblocks + 3) / 4;
    if (gx > 64) gx = 64;
    for (int64_t p0 = 0; p0 < planes; p0 += 65535) {
        const int64_t np = planes - p0 < 65535 ? planes - p0 : 65535;
        hipLaunchKernelGGL((pack_masks_kernel<T>), dim3(gx, (unsigned)np), dim3(256), 0, stream, masks + p0 * plane_stride,
                           HW, plane_stride, packed + p0 * packed_stride, packed_stride);
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
