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
//  * Two-phase form on FIXED SLOTS (the per-frame step of the evaluator without host round trips,
//    model_encoder.py:115-134 + boxlist_ops.py:15-29): proposal_boxes_kernel evaluates the pasted values of every RAW
//    proposal only for its tight box (no plane is written), nms_slots_kernel ranks / suppresses / truncates per image
//    into K slots, paste_kept_kernel then pastes ONLY the kept proposals straight into planes [images, K, H*W] (+ 1-bit
//    planes, kept boxes / scores, the ROIAlign roi rows) in score order.  Same arithmetic (paste_value below), so the
//    planes and boxes are bit identical to paste-everything + NMS + gather -- without the gather copy, with the kept
//    counts left on the device (n_valid of the matching entry points).  Raw inputs may be CLIP RESIDENT ([T, images,
//    R, ...]) with the frame index read from a device scalar (`step`): a captured HIP graph then replays the whole
//    frame step with no host input.
//  * nms_kernel: filter_results' NMS + top-k (dmm/utils/boxlist_ops.py:15-29; maskrcnn_benchmark nms semantics:
//    descending score, legacy +1 areas, IoU > thresh suppresses).  One workgroup per image, <= 1024 boxes: rank by
//    counting (stable), pairwise suppression bitmask in LDS, serial greedy scan by one lane.
#include <stdlib.h>

#include "dmm_common.h"

namespace dmm {

constexpr int kNmsMax = 1024;

constexpr int kPasteIters = 4;       // 1024-pixel steps per workgroup (a band of the plane)

// Geometry of one paste (masker.py:110-150): pad the M x M probabilities, expand the box about its centre by
// (M + 2 pad) / M, truncate to int, clip to the image.  paste_value = one pixel of F.interpolate(bilinear,
// align_corners=False) of the padded mask resized to the box -- every kernel that needs a pasted value calls THIS, so
// planes, tight boxes and 1-bit planes agree bit for bit whichever kernel produced them.
struct PasteGeom {
    int Mp, bx0, by0, x_0, y_0, x_1, y_1;
    float sh, sw;
    __device__ __forceinline__ bool inside(int y, int x) const { return y >= y_0 && y < y_1 && x >= x_0 && x < x_1; }
};
__device__ __forceinline__ PasteGeom paste_geom(const float *__restrict__ box, int M, int padding, int im_h, int im_w) {
    PasteGeom g;
    g.Mp = M + 2 * padding;
    const float scale = (float)((double)g.Mp / (double)M);
    float w_half = (box[2] - box[0]) * 0.5f, h_half = (box[3] - box[1]) * 0.5f;
    const float x_c = (box[2] + box[0]) * 0.5f, y_c = (box[3] + box[1]) * 0.5f;
    w_half = w_half * scale;
    h_half = h_half * scale;
    g.bx0 = (int)(x_c - w_half);
    g.by0 = (int)(y_c - h_half);
    const int bx1 = (int)(x_c + w_half), by1 = (int)(y_c + h_half);
    int w = bx1 - g.bx0 + 1, h = by1 - g.by0 + 1;
    w = w < 1 ? 1 : w;
    h = h < 1 ? 1 : h;
    g.x_0 = max(g.bx0, 0);
    g.y_0 = max(g.by0, 0);
    g.x_1 = min(bx1 + 1, im_w);
    g.y_1 = min(by1 + 1, im_h);
    g.sh = (float)g.Mp / (float)h;
    g.sw = (float)g.Mp / (float)w;
    return g;
}
__device__ __forceinline__ float paste_value(const PasteGeom &g, const float *pad_s, int y, int x) {
    const int Mp = g.Mp;
    float ry = __builtin_fmaf(g.sh, (float)(y - g.by0) + 0.5f, -0.5f);
    ry = ry < 0.0f ? 0.0f : ry;
    const int iy0 = (int)ry, iy1 = iy0 + (iy0 < Mp - 1 ? 1 : 0);
    const float ly1 = ry - (float)iy0, ly0 = 1.0f - ly1;
    float rx = __builtin_fmaf(g.sw, (float)(x - g.bx0) + 0.5f, -0.5f);
    rx = rx < 0.0f ? 0.0f : rx;
    const int ix0 = (int)rx, ix1 = ix0 + (ix0 < Mp - 1 ? 1 : 0);
    const float lx1 = rx - (float)ix0, lx0 = 1.0f - lx1;
    const float t1 = lx1 * pad_s[iy0 * Mp + ix1], b1 = lx1 * pad_s[iy1 * Mp + ix1];
    const float top = __builtin_fmaf(lx0, pad_s[iy0 * Mp + ix0], t1);
    const float bot = __builtin_fmaf(lx0, pad_s[iy1 * Mp + ix0], b1);
    const float lb = ly1 * bot;
    return __builtin_fmaf(ly0, top, lb);
}
__device__ __forceinline__ void stage_padded(const float *__restrict__ prob_p, int M, int padding, float *pad_s) {
    const int Mp = M + 2 * padding;
    for (int i = threadIdx.x; i < Mp * Mp; i += blockDim.x) {
        const int y = i / Mp - padding, x = i % Mp - padding;
        pad_s[i] = (y >= 0 && y < M && x >= 0 && x < M) ? prob_p[y * M + x] : 0.0f;
    }
}

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
    stage_padded(prob + (int64_t)p * M * M, M, padding, pad_s);
    if (threadIdx.x == 0) { box_s[0] = im_w; box_s[1] = im_h; box_s[2] = -1; box_s[3] = -1; }
    __syncthreads();
    const PasteGeom g = paste_geom(boxes + (int64_t)p * 4, M, padding, im_h, im_w);
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
            if (i4 + k < HW && g.inside(y, x)) {
                v = paste_value(g, pad_s, y, x);
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
    // one LDS atomic per wave and coordinate (64 lanes on one LDS address serialise: 4096 of them cost more than the
    // whole evaluation loop)
    xmin = wave_min_i32(xmin);
    ymin = wave_min_i32(ymin);
    xmax = -wave_min_i32(-xmax);
    ymax = -wave_min_i32(-ymax);
    if ((threadIdx.x & 63) == 0 && xmax >= 0) {
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

// NMS + top-k of one image by one workgroup (256 threads): bx [n,4], sc [n] -> keep_out[0 .. cnt) = kept local indices in
// descending score order, *count_out = cnt.  order_s [kNmsMax], supp_s [kNmsMax * kNmsMax / 32] are the caller's LDS.
__device__ __forceinline__ void nms_image(const float *__restrict__ bx, const float *__restrict__ sc, int n, float thresh,
                                          int max_keep, int32_t *__restrict__ keep_out, int32_t *__restrict__ count_out,
                                          int *order_s, unsigned *supp_s) {
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
            keep_out[cnt] = order_s[a];
            ++cnt;
            if (max_keep > 0 && cnt >= max_keep) break;
            for (int wd = a >> 5; wd < words; ++wd) dead[wd] |= supp_s[a * (kNmsMax / 32) + wd];
        }
        *count_out = cnt;
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
    nms_image(boxes + (int64_t)beg * 4, scores + beg, n, thresh, max_keep, keep + beg, keep_count + img, order_s, supp_s);
}

// ---- two-phase form on fixed slots -------------------------------------------------------------------------------
// Raw proposals of `images` images, R slots each: prob [.., images, R, M, M], boxes [.., images, R, 4], scores / counts
// [.., images, R] / [.., images]; the leading axis is the frame of a clip, selected by the DEVICE scalar *step (NULL = 0).
__device__ __forceinline__ int64_t step_of(const int32_t *__restrict__ step) { return step ? (int64_t)step[0] : 0; }

// Phase 1.  grid = images * R; block = 1024: tight box of (pasted value > thresh) of one raw proposal, by evaluating the
// pasted values inside its (clipped) box only -- nothing outside can pass (the plane is zero there and the reference
// tests v > thresh on the pasted region's values only through the same branch, paste_masks_kernel above).
__global__ __launch_bounds__(1024) void proposal_boxes_kernel(const float *__restrict__ prob, const float *__restrict__ boxes,
                                                             const int32_t *__restrict__ counts, int images, int R, int M,
                                                             int im_h, int im_w, float thresh, int padding,
                                                             const int32_t *__restrict__ step,
                                                             float *__restrict__ tight /*[images,R,4]*/) {
    __shared__ float pad_s[64 * 64];
    __shared__ int box_s[4];
    const int img = blockIdx.x / R, r = blockIdx.x - img * R;
    const int64_t fi = step_of(step) * images + img;
    float *o = tight + (int64_t)blockIdx.x * 4;
    if (counts && r >= counts[fi]) {                                  // empty raw slot
        if (threadIdx.x < 4) o[threadIdx.x] = 0.0f;
        return;
    }
    const int64_t p = fi * R + r;
    stage_padded(prob + p * M * M, M, padding, pad_s);
    if (threadIdx.x == 0) { box_s[0] = im_w; box_s[1] = im_h; box_s[2] = -1; box_s[3] = -1; }
    __syncthreads();
    const PasteGeom g = paste_geom(boxes + p * 4, M, padding, im_h, im_w);
    const int rw = g.x_1 - g.x_0, rh = g.y_1 - g.y_0;
    int xmin = im_w, ymin = im_h, xmax = -1, ymax = -1;
    if (rw > 0 && rh > 0) {
        for (int i = threadIdx.x; i < rw * rh; i += 1024) {
            const int yy = i / rw, y = g.y_0 + yy, x = g.x_0 + (i - yy * rw);
            if (paste_value(g, pad_s, y, x) > thresh) {
                xmin = min(xmin, x); xmax = max(xmax, x);
                ymin = min(ymin, y); ymax = max(ymax, y);
            }
        }
    }
    // one LDS atomic per wave and coordinate (64 lanes on one LDS address serialise: 4096 of them cost more than the
    // whole evaluation loop)
    xmin = wave_min_i32(xmin);
    ymin = wave_min_i32(ymin);
    xmax = -wave_min_i32(-xmax);
    ymax = -wave_min_i32(-ymax);
    if ((threadIdx.x & 63) == 0 && xmax >= 0) {
        atomicMin(&box_s[0], xmin);
        atomicMin(&box_s[1], ymin);
        atomicMax(&box_s[2], xmax);
        atomicMax(&box_s[3], ymax);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (box_s[2] < 0) { o[0] = 0.0f; o[1] = 0.0f; o[2] = (float)im_h; o[3] = (float)im_w; }       // masker.py:164
        else { o[0] = (float)box_s[0]; o[1] = (float)box_s[1]; o[2] = (float)box_s[2]; o[3] = (float)box_s[3]; }
    }
}

// Phase 2.  grid = images: NMS + top-K on the tight boxes; keep [images, K], keep_count [images].
__global__ __launch_bounds__(256) void nms_slots_kernel(const float *__restrict__ tight, const float *__restrict__ scores,
                                                        const int32_t *__restrict__ counts, int images, int R, float thresh,
                                                        int K, const int32_t *__restrict__ step,
                                                        int32_t *__restrict__ keep, int32_t *__restrict__ keep_count) {
    __shared__ int order_s[kNmsMax];
    __shared__ unsigned supp_s[kNmsMax * (kNmsMax / 32)];
    const int img = blockIdx.x;
    const int64_t fi = step_of(step) * images + img;
    int n = counts ? counts[fi] : R;
    n = n < 0 ? 0 : (n > R ? R : n);
    nms_image(tight + (int64_t)img * R * 4, scores + fi * R, n, thresh, K, keep + (int64_t)img * K, keep_count + img,
              order_s, supp_s);
}


// NMS + top-k of one image with at most 64 boxes by ONE WORKGROUP OF 4 WAVES, no scratch memory: wave 0 ranks the boxes
// by counting (lane i holds box i; scores broadcast with readlane) and leaves them in LDS in rank order; every wave
// then builds a quarter of the suppression matrix -- lane a = ranked box a against ranked boxes c in its 16-column
// slice, each c one broadcast LDS read -- and wave 0 walks the greedy scan over a wave-uniform 64-bit dead mask.  The
// IoU arithmetic is nms_image's, operation for operation, so both give the same kept set.
__device__ __forceinline__ void nms_image_small(const float *__restrict__ bx, const float *__restrict__ sc, int n,
                                                float thresh, int max_keep, int32_t *__restrict__ keep_out,
                                                int32_t *__restrict__ count_out, float *box_s /*[64*4]*/, int *order_s /*[64]*/,
                                                unsigned *row_s /*[64*4]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {
        const bool live = lane < n;
        float4a b = {0.0f, 0.0f, 0.0f, 0.0f};
        float si = 0.0f;
        if (live) {
            b = *reinterpret_cast<const float4a *>(bx + 4 * lane);               // [n,4] rows: 16-byte aligned
            si = sc[lane];
        }
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const float sj = readlane_f32(si, j);
            rank += (sj > si) || (sj == si && j < lane);
        }
        if (live) {
            order_s[rank] = lane;
            *reinterpret_cast<float4a *>(box_s + 4 * rank) = b;
        }
    }
    __syncthreads();
    const int a = lane;
    float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f, r3 = 0.0f;
    if (a < n) {
        const float4a t = *reinterpret_cast<const float4a *>(box_s + 4 * a);
        r0 = t.x; r1 = t.y; r2 = t.z; r3 = t.w;
    }
    const float ai = (r2 - r0 + 1.0f) * (r3 - r1 + 1.0f);
    unsigned bits = 0u;
    const int c_lo = 16 * wave, c_hi = min(n, c_lo + 16);
    for (int c = c_lo; c < c_hi; ++c) {
        const float4a t = *reinterpret_cast<const float4a *>(box_s + 4 * c);    // one address for the wave: broadcast
        const float l = fmaxf(r0, t.x), r = fminf(r2, t.z);
        const float tp = fmaxf(r1, t.y), bt = fminf(r3, t.w);
        const float iw = fmaxf(r - l + 1.0f, 0.0f), ih = fmaxf(bt - tp + 1.0f, 0.0f);
        const float inter = iw * ih;
        const float aj = (t.z - t.x + 1.0f) * (t.w - t.y + 1.0f);
        if (c > a && inter / (ai + aj - inter) > thresh) bits |= 1u << (c - c_lo);
    }
    row_s[4 * a + wave] = bits;                                 // 16 bits per wave
    __syncthreads();
    if (wave == 0) {
        const unsigned long long row = (unsigned long long)row_s[4 * a] | ((unsigned long long)row_s[4 * a + 1] << 16) |
                                       ((unsigned long long)row_s[4 * a + 2] << 32) |
                                       ((unsigned long long)row_s[4 * a + 3] << 48);
        const int src = a < n ? order_s[a] : 0;
        unsigned long long dead = 0ull;
        int cnt = 0;
        for (int k = 0; k < n; ++k) {
            if ((dead >> k) & 1ull) continue;
            const int idx = __builtin_amdgcn_readlane(src, k);
            if (lane == 0) keep_out[cnt] = idx;
            ++cnt;
            if (max_keep > 0 && cnt >= max_keep) break;
            const unsigned lo = __builtin_amdgcn_readlane((unsigned)row, k);
            const unsigned hi = __builtin_amdgcn_readlane((unsigned)(row >> 32), k);
            dead |= ((unsigned long long)hi << 32) | lo;
        }
        if (lane == 0) *count_out = cnt;
    }
}

// grid = images; block = 256: nms_slots_kernel for R <= 64 (the product's 50 raw proposals per frame).
__global__ __launch_bounds__(256) void nms_slots_small_kernel(const float *__restrict__ tight, const float *__restrict__ scores,
                                                             const int32_t *__restrict__ counts, int images, int R,
                                                             float thresh, int K, const int32_t *__restrict__ step,
                                                             int32_t *__restrict__ keep, int32_t *__restrict__ keep_count) {
    __shared__ __attribute__((aligned(16))) float box_s[64 * 4];
    __shared__ int order_s[64];
    __shared__ unsigned row_s[64 * 4];
    const int img = blockIdx.x;
    const int64_t fi = step_of(step) * images + img;
    int n = counts ? counts[fi] : R;
    n = n < 0 ? 0 : (n > R ? R : n);
    nms_image_small(tight + (int64_t)img * R * 4, scores + fi * R, n, thresh, K, keep + (int64_t)img * K, keep_count + img,
                    box_s, order_s, row_s);
}

// Phase 3.  grid = (bands, images * K); block = 256: slot (img, k) receives raw proposal keep[img, k] -- plane, 1-bit
// plane, tight box, score and its ROIAlign row [image index in the feature batch, box]; slots k >= keep_count[img] are
// dead: their plane is left alone (n_valid hides it), score 0, box 0, roi image index -1 (the ROI kernel then writes
// a zero feature row).
__global__ __launch_bounds__(256) void paste_kept_kernel(
    const float *__restrict__ prob, const float *__restrict__ boxes, const float *__restrict__ scores,
    const float *__restrict__ tight, const int32_t *__restrict__ keep, const int32_t *__restrict__ keep_count, int images,
    int R, int M, int K, int im_h, int im_w, int padding, const int32_t *__restrict__ step,
    const int32_t *__restrict__ img_base, float *__restrict__ planes, int64_t plane_stride,
    unsigned long long *__restrict__ packed, int64_t packed_stride, float *__restrict__ kept_boxes,
    float *__restrict__ kept_scores, float *__restrict__ rois) {
    __shared__ float pad_s[64 * 64];
    const int slot = blockIdx.y, img = slot / K, k = slot - img * K;
    const int64_t st = step_of(step), fi = st * images + img;
    const bool live = k < keep_count[img];
    const int r = live ? keep[(int64_t)img * K + k] : 0;
    if (blockIdx.x == 0 && threadIdx.x < 4) {
        const float tb = live ? tight[((int64_t)img * R + r) * 4 + threadIdx.x] : 0.0f;
        if (kept_boxes) kept_boxes[(int64_t)slot * 4 + threadIdx.x] = tb;
        if (rois) rois[(int64_t)slot * 5 + 1 + threadIdx.x] = tb;
        if (threadIdx.x == 0) {
            if (kept_scores) kept_scores[slot] = live ? scores[fi * R + r] : 0.0f;
            if (rois) rois[(int64_t)slot * 5] = live ? (float)((img_base ? img_base[st] : 0) + img) : -1.0f;
        }
    }
    if (!live) return;
    const int64_t p = fi * R + r;
    const PasteGeom g = paste_geom(boxes + p * 4, M, padding, im_h, im_w);
    float *plane = planes ? planes + (int64_t)slot * plane_stride : nullptr;
    const int HW = im_h * im_w;
    const int lane = threadIdx.x & 63;
    const int i_beg = blockIdx.x * kPasteIters * 1024;
    const int i_end = min(((HW + 255) / 256) * 256, (int)(blockIdx.x + 1) * kPasteIters * 1024);
    // a band that lies above or below the (clipped) box is all zeros: no probabilities to stage, nothing to evaluate --
    // three of four bands of a typical proposal
    const int row_a = i_beg / im_w, row_b = (min(i_end, HW) - 1) / im_w;
    if (row_b < g.y_0 || row_a >= g.y_1 || g.x_1 <= g.x_0) {
        for (int i4 = i_beg + 4 * threadIdx.x; i4 < i_end; i4 += 1024) {
            if (plane) {
                if (i4 + 3 < HW) {
                    float4u t;
                    t.x = 0.0f; t.y = 0.0f; t.z = 0.0f; t.w = 0.0f;
                    *reinterpret_cast<float4u *>(plane + i4) = t;
                } else {
                    for (int q = 0; q < 4; ++q)
                        if (i4 + q < HW) plane[i4 + q] = 0.0f;
                }
            }
            if (packed && lane < 4) packed[(int64_t)slot * packed_stride + (i4 - 4 * lane) / 64 + lane] = 0ull;
        }
        return;
    }
    stage_padded(prob + p * M * M, M, padding, pad_s);
    __syncthreads();
    for (int i4 = blockIdx.x * kPasteIters * 1024 + 4 * threadIdx.x; i4 < i_end; i4 += 1024) {
        float vv[4];
        int y = i4 / im_w, x = i4 - y * im_w;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            vv[q] = (i4 + q < HW && g.inside(y, x)) ? paste_value(g, pad_s, y, x) : 0.0f;
            if (++x == im_w) { x = 0; ++y; }
        }
        if (plane) {
            if (i4 + 3 < HW) {
                float4u t;
                t.x = vv[0]; t.y = vv[1]; t.z = vv[2]; t.w = vv[3];
                *reinterpret_cast<float4u *>(plane + i4) = t;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (i4 + q < HW) plane[i4 + q] = vv[q];
            }
        }
        if (packed) {
            const unsigned long long b0 = __ballot(vv[0] > 0.5f), b1 = __ballot(vv[1] > 0.5f);
            const unsigned long long b2 = __ballot(vv[2] > 0.5f), b3 = __ballot(vv[3] > 0.5f);
            if (lane < 4)
                packed[(int64_t)slot * packed_stride + (i4 - 4 * lane) / 64 + lane] =
                    lane == 0 ? b0 : (lane == 1 ? b1 : (lane == 2 ? b2 : b3));
        }
    }
}


// Phase 3 when only the 1-bit planes are wanted (dmm_paste_kept_f32 with planes = NULL: the frame step's epilogue pastes
// the selected proposals on the fly).  grid = (4, images * K); block = 1024 (16 waves): the workgroups of a slot clear
// the slot's words and evaluate only the 256-pixel blocks its (clipped) box touches -- a wave per block, four ballots
// per block, exactly the words paste_kept_kernel would write (the evaluation is the cost: ~14 of 20 us with one
// workgroup per slot, the rest is the chain of dependent index loads).  (The band decomposition above launched 28 workgroups per slot,
// three quarters of them to write zeros: 17.6 us for 4 x 50 slots of 255 x 448; this form: see DESIGN.md.)
__global__ __launch_bounds__(1024) void pack_kept_kernel(
    const float *__restrict__ prob, const float *__restrict__ boxes, const float *__restrict__ scores,
    const float *__restrict__ tight, const int32_t *__restrict__ keep, const int32_t *__restrict__ keep_count, int images,
    int R, int M, int K, int im_h, int im_w, int padding, const int32_t *__restrict__ step,
    const int32_t *__restrict__ img_base, unsigned long long *__restrict__ packed, int64_t packed_stride,
    float *__restrict__ kept_boxes, float *__restrict__ kept_scores, float *__restrict__ rois) {
    __shared__ float pad_s[64 * 64];
    const int slot = blockIdx.y, img = slot / K, k = slot - img * K;
    const int part = blockIdx.x, parts = gridDim.x;             // a slot's blocks are dealt out to `parts` workgroups
    const int64_t st = step_of(step), fi = st * images + img;
    const bool live = k < keep_count[img];
    const int r = live ? keep[(int64_t)img * K + k] : 0;
    if (part == 0 && threadIdx.x < 4) {
        const float tb = live ? tight[((int64_t)img * R + r) * 4 + threadIdx.x] : 0.0f;
        if (kept_boxes) kept_boxes[(int64_t)slot * 4 + threadIdx.x] = tb;
        if (rois) rois[(int64_t)slot * 5 + 1 + threadIdx.x] = tb;
        if (threadIdx.x == 0) {
            if (kept_scores) kept_scores[slot] = live ? scores[fi * R + r] : 0.0f;
            if (rois) rois[(int64_t)slot * 5] = live ? (float)((img_base ? img_base[st] : 0) + img) : -1.0f;
        }
    }
    if (!live) return;
    const int64_t p = fi * R + r;
    const PasteGeom g = paste_geom(boxes + p * 4, M, padding, im_h, im_w);
    const int HW = im_h * im_w, nblk = (HW + 255) / 256;
    unsigned long long *words = packed + (int64_t)slot * packed_stride;
    // blocks of 256 pixels the box's rows touch: [b_lo, b_hi]
    int b_lo = nblk, b_hi = -1;
    if (g.y_1 > g.y_0 && g.x_1 > g.x_0) {
        b_lo = (g.y_0 * im_w + g.x_0) / 256;
        b_hi = ((g.y_1 - 1) * im_w + g.x_1 - 1) / 256;
    }
    for (int w = part * 1024 + threadIdx.x; w < 4 * nblk; w += 1024 * parts)     // everything outside: zero words
        if ((w >> 2) < b_lo || (w >> 2) > b_hi) words[w] = 0ull;
    if (b_hi < b_lo) return;
    stage_padded(prob + p * M * M, M, padding, pad_s);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int q = b_lo + part * 16 + wave; q <= b_hi; q += 16 * parts) {
        const int i4 = 256 * q + 4 * lane;
        int y = i4 / im_w, x = i4 - y * im_w;
        float vv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            vv[e] = (i4 + e < HW && g.inside(y, x)) ? paste_value(g, pad_s, y, x) : 0.0f;
            if (++x == im_w) { x = 0; ++y; }
        }
        const unsigned long long b0 = __ballot(vv[0] > 0.5f), b1 = __ballot(vv[1] > 0.5f);
        const unsigned long long b2 = __ballot(vv[2] > 0.5f), b3 = __ballot(vv[3] > 0.5f);
        if (lane < 4) words[4 * q + lane] = lane == 0 ? b0 : (lane == 1 ? b1 : (lane == 2 ? b2 : b3));
    }
}

// ---- frame-step epilogue on fixed slots -----------------------------------------------------------------------------
// What the evaluator does with the assignment of a frame (match_model.py:134-144 mask mix, dmm_model.py:66-69 / :78-80
// out_mask_last, evaluator.py:134-139 label map) in ONE pass over the pixels, WITHOUT the proposals' soft planes ever
// having been written: full_outmask[m, x] = sum_n Rb[m, n] * plane_n[x] needs plane values of the few selected
// proposals only (test mode: the row maxima), and a plane value is paste_value() of the raw 28 x 28 probabilities -- so
// the kept proposals are pasted ON THE FLY for the selected (row, proposal) pairs, the result goes to full_outmask,
// to the template history (unless the video was skipped), to the label map, and -- thresholded -- to the history's
// 1-bit planes the next frame's cost pass counts on.  Bytes of a frame step: 9 MB out instead of 91 MB of pasted planes
// + three more passes (pack templates, commit, merge).
// Accumulation per row = mask_mix_rows_kernel's: fma(w, v, acc) over the non-zero weights in ascending column order
// from 0 (a single product in test mode), zero weights skipped, the same paste_value -> bit identical to paste + mix.
// grid = (ceil(HW / (1024 kFinishIters)), B); block = 256; thread = 4 consecutive pixels per 1024-pixel step over all
// M <= 8 rows (kFinishIters = 1 measured best: 24 us per 4 x 255x448 step against 31 us with 4 steps per workgroup).  Mp = M_mask + 2 pad <= 32.
constexpr int kFinishRows = 8, kFinishChunk = 16, kFinishPad = 32 * 32, kFinishIters = 1;

__global__ __launch_bounds__(256) void step_finish_kernel(
    const float *__restrict__ Rb, int Pp, const float *__restrict__ prob, const float *__restrict__ boxes,
    const int32_t *__restrict__ keep, const int32_t *__restrict__ keep_count, int B, int R, int Mm, int K, int M, int im_h,
    int im_w, int padding, const int32_t *__restrict__ step, const int32_t *__restrict__ m_valid,
    const int32_t *__restrict__ commit, const int32_t *__restrict__ o_valid, float *__restrict__ full,
    float *__restrict__ hist, unsigned long long *__restrict__ packed_hist, int64_t words,
    uint8_t *__restrict__ labels) {
    __shared__ float pad_s[kFinishChunk][kFinishPad];
    __shared__ PasteGeom geom_s[kFinishChunk];
    __shared__ float w_s[kFinishRows][kFinishChunk];
    __shared__ int uslot_s[DMM_MAX_PROPOSALS];
    __shared__ int wcnt_s[4];
    __shared__ int hit_s[kFinishChunk];
    const int b = blockIdx.y, HW = im_h * im_w;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int Nb = keep_count[b];
    Nb = Nb < 0 ? 0 : (Nb > K ? K : Nb);
    int Mb = m_valid ? m_valid[b] : M;
    Mb = Mb < 0 ? 0 : (Mb > M ? M : Mb);
    if (Nb == 0) Mb = 0;
    const float *Rb_b = Rb + (int64_t)b * M * Pp;
    // the proposals (columns) that carry a non-zero weight in any live row, ascending
    {
        const int n = threadIdx.x;
        bool used = false;
        if (n < Nb)
            for (int m = 0; m < Mb; ++m) used |= Rb_b[(int64_t)m * Pp + n] != 0.0f;
        const unsigned long long bal = __ballot(used);
        if (lane == 0) wcnt_s[wave] = __builtin_popcountll(bal);
        __syncthreads();
        int base = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) base += k < wave ? wcnt_s[k] : 0;
        if (used) uslot_s[base + __builtin_popcountll(bal & ((1ull << lane) - 1ull))] = n;
    }
    __syncthreads();
    const int U = wcnt_s[0] + wcnt_s[1] + wcnt_s[2] + wcnt_s[3];
    const int64_t fi = step_of(step) * B + b;
    const int Mp = Mm + 2 * padding;
    const bool do_commit = commit ? commit[b] != 0 : false;
    int Ob = o_valid ? o_valid[b] : M;
    Ob = Ob < 0 ? 0 : (Ob > M ? M : Ob);
    // this workgroup's pixels: kFinishIters steps of 1024; the image rows they touch decide which proposals matter here
    const int i_beg = blockIdx.x * kFinishIters * 1024;
    const int i_end = min(((HW + 255) / 256) * 256, i_beg + kFinishIters * 1024);
    const int row_a = i_beg / im_w, row_b = (min(i_end, HW) - 1) / im_w;
    int staged = -1;                                            // chunk of used proposals currently in LDS
    for (int it = 0; it < kFinishIters; ++it) {
        const int i4 = i_beg + it * 1024 + 4 * threadIdx.x;
        if (i_beg + it * 1024 >= i_end) break;                  // (uniform)
        int py[4], px[4];
        {
            int y = i4 / im_w, x = i4 - y * im_w;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                py[q] = y; px[q] = x;
                if (++x == im_w) { x = 0; ++y; }
            }
        }
        float acc[kFinishRows][4];
#pragma unroll
        for (int m = 0; m < kFinishRows; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[m][q] = 0.0f;
        for (int c0 = 0; c0 < U; c0 += kFinishChunk) {
            const int nu = min(kFinishChunk, U - c0);
            if (staged != c0) {                                 // (uniform; with <= 16 used proposals: once per workgroup)
                __syncthreads();                                // the previous chunk is consumed
                if (threadIdx.x < nu) {
                    const int64_t p = fi * R + keep[(int64_t)b * K + uslot_s[c0 + threadIdx.x]];
                    const PasteGeom g = paste_geom(boxes + p * 4, Mm, padding, im_h, im_w);
                    geom_s[threadIdx.x] = g;
                    hit_s[threadIdx.x] = (row_b >= g.y_0 && row_a < g.y_1 && g.x_1 > g.x_0) ? 1 : 0;
                }
                if (threadIdx.x < kFinishRows * kFinishChunk) {
                    const int m = threadIdx.x / kFinishChunk, u = threadIdx.x - m * kFinishChunk;
                    w_s[m][u] = (m < Mb && u < nu) ? Rb_b[(int64_t)m * Pp + uslot_s[c0 + u]] : 0.0f;
                }
                __syncthreads();
                for (int t = threadIdx.x; t < nu * Mp * Mp; t += 256) {
                    const int u = t / (Mp * Mp), e = t - u * Mp * Mp;
                    if (!hit_s[u]) continue;                    // its box misses this workgroup's rows: never evaluated
                    const int yy = e / Mp - padding, xx = e % Mp - padding;
                    const int64_t p = fi * R + keep[(int64_t)b * K + uslot_s[c0 + u]];
                    pad_s[u][e] = (yy >= 0 && yy < Mm && xx >= 0 && xx < Mm) ? prob[p * Mm * Mm + yy * Mm + xx] : 0.0f;
                }
                __syncthreads();
                staged = c0;
            }
            for (int u = 0; u < nu; ++u) {
                if (!hit_s[u]) continue;                        // plane is zero here: fma(w, 0, acc) = acc
                const PasteGeom g = geom_s[u];
                float v[4];
                bool any = false;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool in = i4 + q < HW && g.inside(py[q], px[q]);
                    v[q] = in ? paste_value(g, pad_s[u], py[q], px[q]) : 0.0f;
                    any |= in;
                }
                if (__ballot(any) == 0ull) continue;
#pragma unroll
                for (int m = 0; m < kFinishRows; ++m) {
                    const float w = w_s[m][u];
                    if (w != 0.0f) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[m][q] = __builtin_fmaf(w, v[q], acc[m][q]);
                    }
                }
            }
        }
        // ---- outputs of this step ----
        float best[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        int arg[4] = {0, 0, 0, 0};
        const bool whole = i4 + 3 < HW;
#pragma unroll
        for (int m = 0; m < kFinishRows; ++m) {
            if (m >= M) break;
            float *fo = full + ((int64_t)b * M + m) * HW + i4;
            float *ho = hist + ((int64_t)b * M + m) * HW + i4;
            if (whole) {
                float4u t;
                t.x = acc[m][0]; t.y = acc[m][1]; t.z = acc[m][2]; t.w = acc[m][3];
                *reinterpret_cast<float4u *>(fo) = t;
                if (do_commit) *reinterpret_cast<float4u *>(ho) = t;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (i4 + q < HW) {
                        fo[q] = acc[m][q];
                        if (do_commit) ho[q] = acc[m][q];
                    }
            }
            if (m < Ob) {                                       // evaluator.py:134-139: first maximum wins
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (arg[q] == 0 || acc[m][q] > best[q]) { best[q] = acc[m][q]; arg[q] = m + 1; }
            }
            if (do_commit && packed_hist) {                     // the history's 1-bit planes (ballot layout, pad bits 0)
                const unsigned long long b0 = __ballot(acc[m][0] > 0.5f), b1 = __ballot(acc[m][1] > 0.5f);
                const unsigned long long b2 = __ballot(acc[m][2] > 0.5f), b3 = __ballot(acc[m][3] > 0.5f);
                if (lane < 4)
                    packed_hist[((int64_t)b * M + m) * words + (i4 - 4 * lane) / 64 + lane] =
                        lane == 0 ? b0 : (lane == 1 ? b1 : (lane == 2 ? b2 : b3));
            }
        }
        if (labels) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (i4 + q < HW) {
                    const float bg = 1.0f - best[q];
                    labels[(int64_t)b * HW + i4 + q] = (uint8_t)((Ob > 0 && best[q] > bg) ? arg[q] : 0);
                }
        }
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

// ---- two-phase proposal preparation on fixed slots (include/dmm_match.h (7b)) ----
extern "C" int dmm_proposal_boxes_f32(const float *prob, const float *boxes, const int32_t *counts, int images, int R,
                                      int M, int im_h, int im_w, float thresh, int padding, const int32_t *step,
                                      float *tight, dmm_stream_t stream) {
    if (images < 0 || R < 0 || M <= 0 || im_h < 0 || im_w < 0 || padding < 0) return DMM_ERR_BAD_ARG;
    if (images == 0 || R == 0) return DMM_OK;
    if (!prob || !boxes || !tight) return DMM_ERR_BAD_ARG;
    if (M + 2 * padding > 64) return DMM_ERR_UNSUPPORTED;
    if ((int64_t)images * R > 0x7fffffff) return DMM_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(dmm::proposal_boxes_kernel, dim3(images * R), dim3(1024), 0, (hipStream_t)stream, prob, boxes, counts,
                       images, R, M, im_h, im_w, thresh, padding, step, tight);
    return dmm::check_launch();
}

extern "C" int dmm_nms_slots_f32(const float *tight, const float *scores, const int32_t *counts, int images, int R,
                                 float thresh, int K, const int32_t *step, int32_t *keep, int32_t *keep_count,
                                 dmm_stream_t stream) {
    if (images < 0 || R < 0 || K <= 0) return DMM_ERR_BAD_ARG;
    if (images == 0) return DMM_OK;
    if (!tight || !scores || !keep || !keep_count) return DMM_ERR_BAD_ARG;
    if (R > dmm::kNmsMax) return DMM_ERR_UNSUPPORTED;
    const bool no_wave = dmm::opt(DMM_OPT_NMS_WAVE) == 0;
    if (R <= 64 && !no_wave)
        hipLaunchKernelGGL(dmm::nms_slots_small_kernel, dim3(images), dim3(256), 0, (hipStream_t)stream, tight, scores,
                           counts, images, R, thresh, K, step, keep, keep_count);
    else
        hipLaunchKernelGGL(dmm::nms_slots_kernel, dim3(images), dim3(256), 0, (hipStream_t)stream, tight, scores, counts,
                           images, R, thresh, K, step, keep, keep_count);
    return dmm::check_launch();
}

extern "C" int dmm_paste_kept_f32(const float *prob, const float *boxes, const float *scores, const float *tight,
                                  const int32_t *keep, const int32_t *keep_count, int images, int R, int M, int K, int im_h,
                                  int im_w, int padding, const int32_t *step, const int32_t *img_base, float *planes,
                                  int64_t plane_stride, uint64_t *packed, float *kept_boxes, float *kept_scores,
                                  float *rois, dmm_stream_t stream) {
    if (images < 0 || R < 0 || M <= 0 || K <= 0 || im_h < 0 || im_w < 0 || padding < 0) return DMM_ERR_BAD_ARG;
    if (images == 0) return DMM_OK;
    if (!prob || !boxes || !scores || !tight || !keep || !keep_count || (!planes && !packed) ||
        (planes && plane_stride < (int64_t)im_h * im_w))
        return DMM_ERR_BAD_ARG;
    if (M + 2 * padding > 64) return DMM_ERR_UNSUPPORTED;
    if ((int64_t)images * K > 65535) return DMM_ERR_UNSUPPORTED;                  // grid.y
    if (!planes) {                                              // 1-bit planes only: one workgroup per slot
        hipLaunchKernelGGL(dmm::pack_kept_kernel, dim3(4, images * K), dim3(1024), 0, (hipStream_t)stream, prob, boxes, scores,
                           tight, keep, keep_count, images, R, M, K, im_h, im_w, padding, step, img_base,
                           reinterpret_cast<unsigned long long *>(packed), dmm_pack_words(im_h * im_w), kept_boxes,
                           kept_scores, rois);
        return dmm::check_launch();
    }
    const int nsteps = (im_h * im_w + 1023) / 1024;
    int bands = (nsteps + dmm::kPasteIters - 1) / dmm::kPasteIters;
    bands = bands < 1 ? 1 : bands;
    hipLaunchKernelGGL(dmm::paste_kept_kernel, dim3(bands, images * K), dim3(256), 0, (hipStream_t)stream, prob, boxes,
                       scores, tight, keep, keep_count, images, R, M, K, im_h, im_w, padding, step, img_base, planes,
                       plane_stride, reinterpret_cast<unsigned long long *>(packed), dmm_pack_words(im_h * im_w),
                       kept_boxes, kept_scores, rois);
    return dmm::check_launch();
}

extern "C" int dmm_step_finish_f32(const float *Rb, int Pp, const float *prob, const float *boxes, const int32_t *keep,
                                   const int32_t *keep_count, int B, int R, int Mm, int K, int M, int im_h, int im_w,
                                   int padding, const int32_t *step, const int32_t *m_valid, const int32_t *commit,
                                   const int32_t *o_valid, float *full, float *hist, uint64_t *packed_hist,
                                   uint8_t *labels, dmm_stream_t stream) {
    if (B < 0 || R < 0 || Mm <= 0 || K <= 0 || M < 0 || im_h < 0 || im_w < 0 || padding < 0) return DMM_ERR_BAD_ARG;
    if (B == 0 || M == 0 || im_h * im_w == 0) return DMM_OK;
    if (!Rb || !prob || !boxes || !keep || !keep_count || !full || !hist) return DMM_ERR_BAD_ARG;
    if (M > dmm::kFinishRows || Mm + 2 * padding > 32 || Pp > DMM_MAX_PROPOSALS || K > Pp || B > 65535)
        return DMM_ERR_UNSUPPORTED;
    const int HW = im_h * im_w;
    hipLaunchKernelGGL(dmm::step_finish_kernel, dim3((HW + 1024 * dmm::kFinishIters - 1) / (1024 * dmm::kFinishIters), B), dim3(256), 0, (hipStream_t)stream, Rb, Pp,
                       prob, boxes, keep, keep_count, B, R, Mm, K, M, im_h, im_w, padding, step, m_valid, commit, o_valid,
                       full, hist, reinterpret_cast<unsigned long long *>(packed_hist), dmm_pack_words(HW), labels);
    return dmm::check_launch();
}
