// dmm_video.hip -- the two mask reductions of the per-video frame loop around the matching layer
// (the step right after the path), on gfx950.
//
//  * mask_boxes_kernel: ohw_mask2boxlist of the reference (dmm/utils/utils.py:179-210 with
//    binmask_to_bbox_xyxy_pt :114-143): tight xyxy box of (mask > thresh) for every template plane, the
//    whole frame [0, 0, W-1, H-1] when the plane is empty, and template_valid = plane has a positive pixel
//    (reference: plane sum > 0; identical for the non-negative masks the loop produces).  The reference
//    does a nonzero() + 4 host .item() syncs per object; here one workgroup per plane streams it once.
//  * merge_labels_kernel: the label map written per frame (dmm/modules/evaluator.py:134-139):
//    bg = 1 - max_o m[o];  label = argmax([bg, m[0], ..., m[O_b-1]]) with the FIRST maximum winning (torch CPU
//    max(dim) keeps the earlier index on ties).  The reference runs max / 1- / cat / max again (4 passes
//    + a [O+1,H,W] copy); here every plane is read once and one byte per pixel is written.
//
// Roofline: HBM (reads of the O template planes).  Algorithmic bytes: O*HW*4 (+ HW for the labels).
#include "dmm_common.h"

namespace dmm {

// grid = R planes; block = 256.  Each thread takes 4 consecutive pixels per step.
__global__ __launch_bounds__(256) void mask_boxes_kernel(const float *__restrict__ masks, int64_t stride, int H, int W,
                                                         float thresh, float *__restrict__ boxes,
                                                         int32_t *__restrict__ valid) {
    __shared__ int red_s[4][5];
    const float *plane = masks + (int64_t)blockIdx.x * stride;
    const int HW = H * W;
    int xmin = W, ymin = H, xmax = -1, ymax = -1, pos = 0;
    for (int i4 = 4 * threadIdx.x; i4 < HW; i4 += 1024) {
        float v[4];
        if (i4 + 3 < HW) {
            MaskIO<float>::load4(plane + i4, v);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = i4 + k < HW ? plane[i4 + k] : 0.0f;
        }
        int y = i4 / W, x = i4 - y * W;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i4 + k < HW) {
                if (v[k] > thresh) {
                    xmin = min(xmin, x); xmax = max(xmax, x);
                    ymin = min(ymin, y); ymax = max(ymax, y);
                }
                pos |= v[k] > 0.0f;
            }
            if (++x == W) { x = 0; ++y; }
        }
    }
    xmin = wave_min_i32(xmin);
    ymin = wave_min_i32(ymin);
    xmax = -wave_min_i32(-xmax);
    ymax = -wave_min_i32(-ymax);
    pos = __ballot(pos != 0) != 0ull;
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red_s[wave][0] = xmin; red_s[wave][1] = ymin; red_s[wave][2] = xmax; red_s[wave][3] = ymax; red_s[wave][4] = pos;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            xmin = min(xmin, red_s[w][0]); ymin = min(ymin, red_s[w][1]);
            xmax = max(xmax, red_s[w][2]); ymax = max(ymax, red_s[w][3]);
            pos |= red_s[w][4];
        }
        float *o = boxes + (int64_t)blockIdx.x * 4;
        if (xmax < 0) { o[0] = 0.0f; o[1] = 0.0f; o[2] = (float)(W - 1); o[3] = (float)(H - 1); }
        else { o[0] = (float)xmin; o[1] = (float)ymin; o[2] = (float)xmax; o[3] = (float)ymax; }
        valid[blockIdx.x] = pos;
    }
}

// grid = (pixel blocks of 1024, B); block = 256; thread = 4 consecutive pixels over all live planes.
__global__ __launch_bounds__(256) void merge_labels_kernel(const float *__restrict__ masks, int O, int HW, int64_t s_b,
                                                           int64_t s_o, const int32_t *__restrict__ o_valid,
                                                           uint8_t *__restrict__ labels) {
    const int b = blockIdx.y;
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x >= HW) return;
    int Ob = o_valid ? o_valid[b] : O;
    Ob = Ob < 0 ? 0 : (Ob > O ? O : Ob);
    const float *base = masks + (int64_t)b * s_b + x;
    const bool full = x + 3 < HW;
    float best[4];
    int arg[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { best[k] = 0.0f; arg[k] = 0; }
    // first pass in registers: running first-occurrence maximum over the object planes (arg = o + 1)
    for (int o0 = 0; o0 < Ob; o0 += 4) {
        float v[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int o = o0 + u < Ob ? o0 + u : Ob - 1;
            const float *p = base + (int64_t)o * s_o;
            if (full) {
                MaskIO<float>::load4(p, v[u]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[u][k] = x + k < HW ? p[k] : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (o0 + u < Ob) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (arg[k] == 0 || v[u][k] > best[k]) { best[k] = v[u][k]; arg[k] = o0 + u + 1; }
            }
    }
    uint8_t *out = labels + (int64_t)b * HW + x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (x + k >= HW) break;
        // background score 1 - max sits at index 0 and wins ties
        const float bg = 1.0f - best[k];
        out[k] = (uint8_t)((Ob > 0 && best[k] > bg) ? arg[k] : 0);
    }
}

}  // namespace dmm

extern "C" int dmm_mask_boxes_f32(const float *masks, int R, int H, int W, int64_t plane_stride, float thresh,
                                  float *boxes, int32_t *valid, dmm_stream_t stream) {
    if (R < 0 || H <= 0 || W <= 0) return DMM_ERR_BAD_ARG;
    if (R == 0) return DMM_OK;
    if (!masks || !boxes || !valid || plane_stride < (int64_t)H * W) return DMM_ERR_BAD_ARG;
    hipLaunchKernelGGL(dmm::mask_boxes_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, masks, plane_stride, H, W,
                       thresh, boxes, valid);
    return dmm::check_launch();
}

extern "C" int dmm_merge_labels_f32(const float *masks, int B, int O, int HW, int64_t stride_b, int64_t stride_o,
                                    const int32_t *o_valid, uint8_t *labels, dmm_stream_t stream) {
    if (B < 0 || O < 0 || HW < 0 || O > 255) return DMM_ERR_BAD_ARG;
    if (B == 0 || HW == 0) return DMM_OK;
    if (!labels || (O > 0 && !masks)) return DMM_ERR_BAD_ARG;
    if (B > 65535) return DMM_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(dmm::merge_labels_kernel, dim3((HW + 1023) / 1024, B), dim3(256), 0, (hipStream_t)stream, masks, O,
                       HW, stride_b, stride_o, o_valid, labels);
    return dmm::check_launch();
}

// ---------------------------------------------------------------------------------------------
// ragged_pad_kernel: out[b, i, :] = src_b[i, :] for i < counts[b], zeros up to P_max -- the batching step of the
// per-video driver (reference: one MatchModel call per video, dmm/modules/dmm_model.py:62-82; here the videos of a step
// run as one ragged launch and their per-video feature / score / packed-plane blocks are stacked first).  One launch
// for the whole batch instead of a zero fill + one copy per video (12 launches per frame step in the frame loop).
// Rows are moved as dwords; src_b = table[b] (device pointer table, like the *_frames entries).
// ---------------------------------------------------------------------------------------------
namespace dmm {
__global__ __launch_bounds__(256) void ragged_pad_kernel(const uint32_t *const *__restrict__ table,
                                                         const int32_t *__restrict__ counts, int P_max,
                                                         int64_t row_dwords, uint32_t *__restrict__ out) {
    const int b = blockIdx.y;
    const int64_t per_frame = (int64_t)P_max * row_dwords;
    const int64_t live = (int64_t)min(counts[b], P_max) * row_dwords;
    typedef const __attribute__((address_space(1))) uint32_t *gptr;
    const gptr src = (gptr)table[b];
    uint32_t *dst = out + (int64_t)b * per_frame;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_frame; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = i < live ? src[i] : 0u;
}
}  // namespace dmm

extern "C" int dmm_ragged_pad(const void *const *src_table, const int32_t *counts, int B, int P_max, int64_t row_bytes,
                              void *out, dmm_stream_t stream) {
    if (B < 0 || P_max < 0 || row_bytes < 0 || (row_bytes & 3)) return DMM_ERR_BAD_ARG;
    if (B == 0 || P_max == 0 || row_bytes == 0) return DMM_OK;
    if (!src_table || !counts || !out) return DMM_ERR_BAD_ARG;
    if (B > 65535) return DMM_ERR_UNSUPPORTED;
    const int64_t per_frame = (int64_t)P_max * (row_bytes / 4);
    int64_t blocks = (per_frame + 1023) / 1024;                 // ~4 dwords per thread
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(dmm::ragged_pad_kernel, dim3((unsigned)blocks, B), dim3(256), 0, (hipStream_t)stream,
                       (const uint32_t *const *)src_table, counts, P_max, row_bytes / 4, (uint32_t *)out);
    return dmm::check_launch();
}

// ---------------------------------------------------------------------------------------------
// Device-resident frame cursor (include/dmm_match.h (8b)).  The evaluator's frame loop (dmm/modules/evaluator.py:63-213)
// advances t on the host and rebuilds every per-frame argument there; with the raw proposals of a clip resident on the
// device and the frame index in a DEVICE scalar, one captured HIP graph replays every frame step of the clip with no
// host input: step_select copies row *step of a per-clip int table (live template counts / commit flags per video) to
// a fixed address, step_advance increments the scalar as the graph's last node.
// commit_masks: out_mask_last of the per-video driver (dmm/modules/dmm_model.py:66-69, :78-80) -- a video's template
// planes are replaced by the matched masks unless the video was skipped this frame (no live template / 'extra' frame),
// then they stay as they were.
// ---------------------------------------------------------------------------------------------
namespace dmm {
__global__ void step_select_kernel(const int32_t *__restrict__ table, const int32_t *__restrict__ step, int n,
                                   int32_t *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = table[(int64_t)step[0] * n + i];
}
__global__ void step_advance_kernel(int32_t *step) { step[0] = step[0] + 1; }

// grid = (blocks, B); block = 256; 16 bytes per thread and step.
__global__ __launch_bounds__(256) void commit_masks_kernel(const float *__restrict__ full, float *__restrict__ hist,
                                                           const int32_t *__restrict__ commit, int64_t per_video) {
    const int b = blockIdx.y;
    if (commit[b] == 0) return;
    const float *src = full + (int64_t)b * per_video;
    float *dst = hist + (int64_t)b * per_video;
    const int64_t n4 = per_video >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float v[4];
        MaskIO<float>::load4(src + 4 * i, v);
        float4u t;
        t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3];
        *reinterpret_cast<float4u *>(dst + 4 * i) = t;
    }
    if (blockIdx.x == 0 && threadIdx.x < (per_video & 3)) dst[4 * n4 + threadIdx.x] = src[4 * n4 + threadIdx.x];
}
}  // namespace dmm

extern "C" int dmm_step_select_i32(const int32_t *table, const int32_t *step, int n, int32_t *out, dmm_stream_t stream) {
    if (n < 0) return DMM_ERR_BAD_ARG;
    if (n == 0) return DMM_OK;
    if (!table || !step || !out) return DMM_ERR_BAD_ARG;
    hipLaunchKernelGGL(dmm::step_select_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, table, step, n,
                       out);
    return dmm::check_launch();
}

extern "C" int dmm_step_advance(int32_t *step, dmm_stream_t stream) {
    if (!step) return DMM_ERR_BAD_ARG;
    hipLaunchKernelGGL(dmm::step_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step);
    return dmm::check_launch();
}

extern "C" int dmm_commit_masks_f32(const float *full, float *hist, const int32_t *commit, int B, int64_t per_video,
                                    dmm_stream_t stream) {
    if (B < 0 || per_video < 0) return DMM_ERR_BAD_ARG;
    if (B == 0 || per_video == 0) return DMM_OK;
    if (!full || !hist || !commit) return DMM_ERR_BAD_ARG;
    if (B > 65535) return DMM_ERR_UNSUPPORTED;
    int64_t blocks = (per_video / 4 + 1023) / 1024;
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(dmm::commit_masks_kernel, dim3((unsigned)blocks, B), dim3(256), 0, (hipStream_t)stream, full, hist,
                       commit, per_video);
    return dmm::check_launch();
}
