// dmm_proposals.hip -- proposal preprocessing on gfx950 (the step right before the path).
//
//  * paste_masks_kernel: paste_mask_in_image + binmask_to_box of the reference (dmm/utils/masker.py:110-173,
//    driven per proposal by Masker.forward_single_image :186-215, a Python loop on the host in the reference):
//    pad the M x M mask probability, expand the box about its centre, truncate to int, bilinear-resize the padded
//    mask into the box (torch F.interpolate, align_corners=False) and paste it into an H x W plane -- exactly the
//    soft [P, H, W] masks the matching layer consumes -- and return the tight box of (plane > thresh).
//    A proposal's plane is cut into bands of 4096 pixels, one workgroup each (a first one-workgroup-per-proposal version
//    kept 50 CUs busy and wrote at 0.27 TB/s); the (M+2)^2 padded mask sits in LDS, 16-byte stores, the whole plane is
//    produced (zeros outside the box) so no separate memset is needed.  HBM bound (plane writes).
//  * nms_kernel: filter_results' NMS + top-k (dmm/utils/boxlist_ops.py:15-29; maskrcnn_benchmark nms semantics:
//    descending score, legacy +1 areas, IoU > thresh suppresses).  One workgroup per image, <= 1024 boxes: rank by
//    counting (stable), pairwise suppression bitmask in LDS, serial greedy scan by one lane.
#include "dmm_common.h"

namespace dmm {

constexpr int kNmsMax = 1024;

constexpr int kPasteIters = 4;       // 1024-pixel steps per workgroup (a band of the plane)

// Tight boxes are reduced ACROSS the band workgroups of a proposal with integer atomics on the bit patterns of the
// (non-negative, integer-valued) float coordinates in new_boxes itself: init = [W, H, -1, -1], atomicMin / atomicMax as
// signed ints (order-preserving for floats >= 0; -1.0f is below all of them), then a finalize pass writes the
// reference's empty-mask box.
__global__ void paste_init_kernel(float *__restrict__ new_boxes, int P, int im_h, int im_w) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    float *nb = new_boxes + (int64_t)p * 4;
    nb[0] = (float)im_w; nb[1] = (float)im_h; nb[2] = -1.0f; nb[3] = -1.0f;
}
__global__ void paste_finalize_kernel(float *__restrict__ new_boxes, int P, int im_h, int im_w) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    float *nb = new_boxes + (int64_t)p * 4;
    if (nb[2] < 0.0f) { nb[0] = 0.0f; nb[1] = 0.0f; nb[2] = (float)im_h; nb[3] = (float)im_w; }   // masker.py:164
}

// grid = (bands, P); block = 256.  Every workgroup re-stages the (M + 2 pad)^2 probabilities in LDS (3.6 KB) and writes
// kPasteIters * 1024 consecutive pixels of the plane, 16 bytes per thread and store.
__global__ __launch_bounds__(256) void paste_masks_kernel(const float *__restrict__ prob, int M,
                                                          const float *__restrict__ boxes, int im_h, int im_w,
                                                          float thresh, int padding, float *__restrict__ planes,
                                                          int64_t plane_stride, float *__restrict__ new_boxes,
                                                          unsigned long long *__restrict__ packed, int64_t packed_stride) {
    __shared__ float pad_s[64 * 64];
    __shared__ int box_s[4];
    const int p = blockIdx.y;
    const int Mp = M + 2 * padding;
    for (int i = threadIdx.x; i < Mp * Mp; i += 256) {
        const int y = i / Mp - padding, x = i % Mp - padding;
        pad_s[i] = (y >= 0 && y < M && x >= 0 && x < M) ? prob[(int64_t)p * M * M + y * M + x] : 0.0f;
    }
    if (threadIdx.x == 0) { box_s[0] = im_w; box_s[1] = im_h; box_s[2] = -1; box_s[3] = -1; }
    __syncthreads();
    const float *box = boxes + (int64_t)p * 4;
    const float scale = (float)((double)Mp / (double)M);
    float w_half = (box[2] - box[0]) * 0.5f, h_half = (box[3] - box[1]) * 0.5f;
    const float x_c = (box[2] + box[0]) * 0.5f, y_c = (box[3] + box[1]) * 0.5f;
    w_half = w_half * scale;
    h_half = h_half * scale;
    const int bx0 = (int)(x_c - w_half), by0 = (int)(y_c - h_half);
    const int bx1 = (int)(x_c + w_half), by1 = (int)(y_c + h_half);
    int w = bx1 - bx0 + 1, h = by1 - by0 + 1;
    w = w < 1 ? 1 : w;
    h = h < 1 ? 1 : h;
    const int x_0 = max(bx0, 0), y_0 = max(by0, 0);
    const int x_1 = min(bx1 + 1, im_w), y_1 = min(by1 + 1, im_h);
    const float sh = (float)Mp / (float)h, sw = (float)Mp / (float)w;
    float *plane = planes + (int64_t)p * plane_stride;
    int xmin = im_w, ymin = im_h, xmax = -1, ymax = -1;
    // each thread produces 4 consecutive pixels per step, a wave 256: exactly one block of the packed ballot layout
    const int HW = im_h * im_w;
    const int lane = threadIdx.x & 63;
    const int i_end = min(((HW + 255) / 256) * 256, (int)(blockIdx.x + 1) * kPasteIters * 1024);
    for (int i4 = blockIdx.x * kPasteIters * 1024 + 4 * threadIdx.x; i4 < i_end; i4 += 1024) {
        float vv[4];
        int y = i4 / im_w, x = i4 - y * im_w;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = 0.0f;
            if (i4 + k < HW && y >= y_0 && y < y_1 && x >= x_0 && x < x_1) {
                float ry = __builtin_fmaf(sh, (float)(y - by0) + 0.5f, -0.5f);
                ry = ry < 0.0f ? 0.0f : ry;
                const int iy0 = (int)ry, iy1 = iy0 + (iy0 < Mp - 1 ? 1 : 0);
                const float ly1 = ry - (float)iy0, ly0 = 1.0f - ly1;
                float rx = __builtin_fmaf(sw, (float)(x - bx0) + 0.5f, -0.5f);
                rx = rx < 0.0f ? 0.0f : rx;
                const int ix0 = (int)rx, ix1 = ix0 + (ix0 < Mp - 1 ? 1 : 0);
                const float lx1 = rx - (float)ix0, lx0 = 1.0f - lx1;
                const float t1 = lx1 * pad_s[iy0 * Mp + ix1], b1 = lx1 * pad_s[iy1 * Mp + ix1];
                const float top = __builtin_fmaf(lx0, pad_s[iy0 * Mp + ix0], t1);
                const float bot = __builtin_fmaf(lx0, pad_s[iy1 * Mp + ix0], b1);
                const float lb = ly1 * bot;
                v = __builtin_fmaf(ly0, top, lb);
                if (v > thresh) {
                    xmin = min(xmin, x); xmax = max(xmax, x);
                    ymin = min(ymin, y); ymax = max(ymax, y);
                }
            }
            vv[k] = v;
            if (++x == im_w) { x = 0; ++y; }
        }
        if (i4 + 3 < HW) {
            float4u t;
            t.x = vv[0]; t.y = vv[1]; t.z = vv[2]; t.w = vv[3];
            *reinterpret_cast<float4u *>(plane + i4) = t;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i4 + k < HW) plane[i4 + k] = vv[k];
        }
        if (packed) {
            const unsigned long long b0 = __ballot(vv[0] > 0.5f), b1 = __ballot(vv[1] > 0.5f);
            const unsigned long long b2 = __ballot(vv[2] > 0.5f), b3 = __ballot(vv[3] > 0.5f);
            if (lane < 4)
                packed[(int64_t)p * packed_stride + (i4 - 4 * lane) / 64 + lane] =
                    lane == 0 ? b0 : (lane == 1 ? b1 : (lane == 2 ? b2 : b3));
        }
    }
    if (xmax >= 0) {
        atomicMin(&box_s[0], xmin);
        atomicMin(&box_s[1], ymin);
        atomicMax(&box_s[2], xmax);
        atomicMax(&box_s[3], ymax);
    }
    __syncthreads();
    if (threadIdx.x == 0 && box_s[2] >= 0) {
        int *nb = reinterpret_cast<int *>(new_boxes + (int64_t)p * 4);
        atomicMin(&nb[0], __float_as_int((float)box_s[0]));
        atomicMin(&nb[1], __float_as_int((float)box_s[1]));
        atomicMax(&nb[2], __float_as_int((float)box_s[2]));
        atomicMax(&nb[3], __float_as_int((float)box_s[3]));
    }
}

// grid = images; boxes [sum n, 4], scores [sum n], offsets [images + 1] (device int32).
__global__ __launch_bounds__(256) void nms_kernel(const float *__restrict__ boxes, const float *__restrict__ scores,
                                                  const int32_t *__restrict__ offsets, float thresh, int max_keep,
                                                  int32_t *__restrict__ keep, int32_t *__restrict__ keep_count) {
    __shared__ int order_s[kNmsMax];
    __shared__ unsigned supp_s[kNmsMax * (kNmsMax / 32)];   // [ranked i][word of ranked j]
    const int img = blockIdx.x;
    const int beg = offsets[img], n = offsets[img + 1] - beg;
    const float *bx = boxes + (int64_t)beg * 4;
    const float *sc = scores + beg;
    const int words = (n + 31) / 32;
    // rank by counting: position = #boxes with a higher score (or equal score and lower index) -> stable
    for (int i = threadIdx.x; i < n; i += 256) {
        int r = 0;
        const float si = sc[i];
        for (int j = 0; j < n; ++j) r += (sc[j] > si) || (sc[j] == si && j < i);
        order_s[r] = i;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < n * words; t += 256) {
        const int a = t / words, wd = t - a * words;
        const float *bi = bx + 4 * order_s[a];
        const float ai = (bi[2] - bi[0] + 1.0f) * (bi[3] - bi[1] + 1.0f);
        unsigned bits = 0;
        for (int k = 0; k < 32; ++k) {
            const int c = wd * 32 + k;
            if (c > a && c < n) {
                const float *bj = bx + 4 * order_s[c];
                const float l = fmaxf(bi[0], bj[0]), r = fminf(bi[2], bj[2]);
                const float tp = fmaxf(bi[1], bj[1]), bt = fminf(bi[3], bj[3]);
                const float iw = fmaxf(r - l + 1.0f, 0.0f), ih = fmaxf(bt - tp + 1.0f, 0.0f);
                const float inter = iw * ih;
                const float aj = (bj[2] - bj[0] + 1.0f) * (bj[3] - bj[1] + 1.0f);
                if (inter / (ai + aj - inter) > thresh) bits |= 1u << k;
            }
        }
        supp_s[a * (kNmsMax / 32) + wd] = bits;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned dead[kNmsMax / 32];
        for (int wd = 0; wd < words; ++wd) dead[wd] = 0;
        int cnt = 0;
        for (int a = 0; a < n; ++a) {
            if (dead[a >> 5] & (1u << (a & 31))) continue;
            keep[beg + cnt] = order_s[a];
            ++cnt;
            if (max_keep > 0 && cnt >= max_keep) break;
            for (int wd = a >> 5; wd < words; ++wd) dead[wd] |= supp_s[a * (kNmsMax / 32) + wd];
        }
        keep_count[img] = cnt;
    }
}

}  // namespace dmm

extern "C" int dmm_paste_masks_f32(const float *prob, int P, int M, const float *boxes, int im_h, int im_w, float thresh,
                                   int padding, float *planes, int64_t plane_stride, float *new_boxes,
                                   uint64_t *packed, dmm_stream_t stream) {
    if (P < 0 || M <= 0 || im_h < 0 || im_w < 0 || padding < 0) return DMM_ERR_BAD_ARG;
    if (P == 0) return DMM_OK;
    if (!prob || !boxes || !planes || !new_boxes || plane_stride < (int64_t)im_h * im_w) return DMM_ERR_BAD_ARG;
    if (M + 2 * padding > 64) return DMM_ERR_UNSUPPORTED;
    const int nsteps = (im_h * im_w + 1023) / 1024;
    const int bands = (nsteps + dmm::kPasteIters - 1) / dmm::kPasteIters;
    hipStream_t s = (hipStream_t)stream;
    for (int p0 = 0; p0 < P; p0 += 65535) {                     // grid.y limit
        const int np = P - p0 < 65535 ? P - p0 : 65535;
        float *nb = new_boxes + (int64_t)p0 * 4;
        hipLaunchKernelGGL(dmm::paste_init_kernel, dim3((np + 255) / 256), dim3(256), 0, s, nb, np, im_h, im_w);
        hipLaunchKernelGGL(dmm::paste_masks_kernel, dim3(bands, np), dim3(256), 0, s, prob + (int64_t)p0 * M * M, M,
                           boxes + (int64_t)p0 * 4, im_h, im_w, thresh, padding, planes + (int64_t)p0 * plane_stride,
                           plane_stride, nb,
                           packed ? reinterpret_cast<unsigned long long *>(packed) + (int64_t)p0 * dmm_pack_words(im_h * im_w)
                                  : nullptr,
                           dmm_pack_words(im_h * im_w));
        hipLaunchKernelGGL(dmm::paste_finalize_kernel, dim3((np + 255) / 256), dim3(256), 0, s, nb, np, im_h, im_w);
    }
    return dmm::check_launch();
}

extern "C" int dmm_nms_f32(const float *boxes, const float *scores, const int32_t *offsets, int images, int max_per_image,
                           float thresh, int max_keep, int32_t *keep, int32_t *keep_count, dmm_stream_t stream) {
    if (images < 0 || max_per_image < 0) return DMM_ERR_BAD_ARG;
    if (images == 0) return DMM_OK;
    if (!boxes || !scores || !offsets || !keep || !keep_count) return DMM_ERR_BAD_ARG;
    if (max_per_image > dmm::kNmsMax) return DMM_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(dmm::nms_kernel, dim3(images), dim3(256), 0, (hipStream_t)stream, boxes, scores, offsets, thresh,
                       max_keep, keep, keep_count);
    return dmm::check_launch();
}
