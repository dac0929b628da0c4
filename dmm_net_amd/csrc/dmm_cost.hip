// dmm_cost.hip -- pairwise binary-mask intersection / area tables on gfx950.
//
// Replaces compute_iou_binary_mask_2D over the expanded [O*P, HW] tensors of the reference
// (dmm/utils/match_helper.py:9-28, called from dmm/modules/match_model.py:83-89 and from
// compute_matching_loss, match_helper.py:34-43).  The reference materialises two O*P*HW fp32
// copies; here every mask plane is read from HBM exactly once.
//
// Roofline: HBM.  Algorithmic bytes per frame = (N+M)*HW*sizeof(elem) (+ 4*M*N table).
//
// Mapping (one workgroup = 4 independent waves, one frame, a contiguous range of 1024-pixel chunks):
//   * a wave takes a 1024-pixel run of a plane as 16-byte loads per lane (4 dwordx4 for fp32, 2 for
//     half / bfloat16; 1 KiB per wave instruction, coalesced), thresholds `> 0.5` and bit-packs
//     through v_cmp -> 64-bit lane masks (the hardware transposer);
//   * the 16 64-bit words are parked in lane `plane` of 32 VGPRs with a lane-select (v_cndmask on
//     lane == plane; proposal n -> lane n of group n/64, template m -> lane m of the template set),
//     so the whole bit tile of a chunk lives in registers -- no LDS, no barriers in the streaming loop;
//   * pair phase: lane = proposal, scalar loop over templates: v_readlane the template word
//     into SGPRs, v_and + v_bcnt accumulate popc(P & T) into acc[m] (registers);
//   * epilogue: the 4 waves fold their integer partials with LDS atomics, then one global
//     atomicAdd per table entry and workgroup (integers: result independent of order).
// VALU work is ~10 % of the HBM time of a chunk; the kernel is a pure stream.
// DMM_PACKED1 input (1 bit per pixel, see dmm_pack.hip) skips the threshold/ballot step: lane p loads the 16 words
// of its plane's chunk straight into the tile.
#include <stdlib.h>

#include <type_traits>

#include "dmm_common.h"
#include "dmm_cosine_lanes.h"

namespace dmm {

constexpr int kChunk = 1024;          // pixels a wave takes from one plane per visit: a contiguous run of
                                      // 4 KiB (fp32) / 2 KiB (16-bit).  A plane row is only 4-byte (2-byte) aligned, so
                                      // every wave load straddles one extra 128-B line; taking the run back to back
                                      // shares those lines (measured HBM over-fetch 8.6 % -> 1.3 % for fp32)
constexpr int kLoadBytes = 8192;      // bytes of plane loads a wave keeps in flight: 2 fp32 planes or 4 16-bit planes
constexpr int kWords = kChunk / 64;   // 64-bit words per plane and chunk (16)
#ifndef DMM_TL_LOAD_BYTES
#define DMM_TL_LOAD_BYTES 8192
#define DMM_TL_MIN_WAVES 4
#endif
constexpr int kCostThreads = 256;     // 4 waves; 1- and 2-wave workgroups measured 2-4 % slower

// One 16-byte load per lane: E = 4 (fp32) or 8 (half / bfloat16) consecutive pixels, kept RAW in 4 VGPRs until the
// ballots consume it; a chunk is kChunk / (64 E) such loads (4 or 2).  Word index of pixel 64*E*j + E*lane + k is
// E*j + k.
template <typename T, bool TAIL>
__device__ __forceinline__ typename MaskIO<T>::Raw load_pixels(const T *plane, int x, int HW) {
    constexpr int E = MaskIO<T>::kVec;
    typename MaskIO<T>::Raw r;
    if (!TAIL || x + E - 1 < HW) {
        r = MaskIO<T>::load_raw(plane + x);
    } else {
        // the chunk that straddles the end of the plane: element-wise, zero (never > 0.5) past the end
        T tmp[E];
#pragma unroll
        for (int k = 0; k < E; ++k) {
            if (x + k < HW) tmp[k] = plane[x + k];
            else __builtin_memset(&tmp[k], 0, sizeof(T));
        }
        __builtin_memcpy(&r, tmp, sizeof(r));
    }
    return r;
}

// Bit tile of one group of <= 64 planes for one chunk: lane p holds the kWords words of plane p.
template <int CH>
struct BitTileT {
    static constexpr int kW = CH / 64;
    int lo[kW], hi[kW];
};
typedef BitTileT<kChunk> BitTile;

// DMM_PACKED1 planes: the words ARE the tile -- lane lane0+p loads the 16 words (128 B) of its plane's chunk.
template <bool TAIL, int CH = kChunk>
__device__ __forceinline__ void fill_tile_packed(BitTileT<CH> &w, const packed_t *base, int64_t plane_stride, int nplanes,
                                                 int x0, int HW, int lane0, bool clear) {
    const int lane = threadIdx.x & 63;
    const int p = lane - lane0;
    const bool mine = p >= 0 && p < nplanes;
    if (clear) {
#pragma unroll
        for (int k = 0; k < (CH / 64); ++k) { w.lo[k] = 0; w.hi[k] = 0; }
    }
    if (mine) {
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(base + (int64_t)p * plane_stride)
                                        + (x0 >> 6);
        const int nwords = 4 * ((HW + 255) / 256);
#pragma unroll
        for (int k = 0; k < (CH / 64); ++k) {
            unsigned long long v = 0;
            if (!TAIL || (x0 >> 6) + k < nwords) v = src[k];
            w.lo[k] = (int)(unsigned)v;
            w.hi[k] = (int)(unsigned)(v >> 32);
        }
    }
}

template <typename T, bool TAIL, int CH = kChunk, int LB = kLoadBytes>
__device__ __forceinline__ void fill_tile(BitTileT<CH> &w, const T *base, int64_t plane_stride, int nplanes,
                                          int x0, int HW, int lane0 = 0, bool clear = true) {
    constexpr int E = MaskIO<T>::kVec;
    constexpr int SUB = CH / (64 * E);
    constexpr int kUnroll = LB / (CH * (int)sizeof(T));
    constexpr int kWords = CH / 64;
    const int lane = threadIdx.x & 63;
    const int x = x0 + lane * E;
    if (clear) {
#pragma unroll
        for (int k = 0; k < kWords; ++k) { w.lo[k] = 0; w.hi[k] = 0; }
    }
    for (int p0 = 0; p0 < nplanes; p0 += kUnroll) {
        typename MaskIO<T>::Raw v[kUnroll][SUB];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            // clamp: planes past the end re-read the last one; their words are never parked
            const int p = p0 + u < nplanes ? p0 + u : nplanes - 1;
#pragma unroll
            for (int j = 0; j < SUB; ++j)
                v[u][j] = load_pixels<T, TAIL>(base + (int64_t)p * plane_stride, x + j * 64 * E, HW);
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int p = p0 + u;
            if (p < nplanes) {
                const bool mine = lane == p + lane0;   // park the words of plane p in lane lane0 + p
#pragma unroll
                for (int j = 0; j < SUB; ++j)
#pragma unroll
                    for (int k = 0; k < E; ++k) {
                        const unsigned long long b = __ballot(MaskIO<T>::gt_half(v[u][j], k));
                        w.lo[E * j + k] = mine ? (int)(unsigned)b : w.lo[E * j + k];
                        w.hi[E * j + k] = mine ? (int)(unsigned)(b >> 32) : w.hi[E * j + k];
                    }
            }
        }
    }
}

template <typename T, bool TAIL, int CH = kChunk, int LB = kLoadBytes>
__device__ __forceinline__ void fill_any(BitTileT<CH> &w, const T *base, int64_t plane_stride, int nplanes, int x0, int HW,
                                         int lane0, bool clear) {
    if constexpr (std::is_same<T, packed_t>::value) fill_tile_packed<TAIL, CH>(w, base, plane_stride, nplanes, x0, HW, lane0, clear);
    else fill_tile<T, TAIL, CH, LB>(w, base, plane_stride, nplanes, x0, HW, lane0, clear);
}

template <typename T, int MT, int NG, bool TAIL, int CH = kChunk, int LB = kLoadBytes>
__device__ __forceinline__ void process_chunk(const T *Pb, const T *Tb, const T *T2b, int64_t sp_n, int64_t st_m,
                                              int64_t st2_m, int Nb, int Mb, int Mrows, int x0, int HW,
                                              unsigned (&acc)[NG][MT], unsigned (&area_p)[NG], unsigned &area_t) {
    // template tile: lanes [0, Mb) = planes of set 1, lanes [Mb, 2*Mb) = planes of set 2 (training: the targets)
    constexpr int kWords = CH / 64;
    BitTileT<CH> tw;
    fill_any<T, TAIL, CH, LB>(tw, Tb, st_m, Mb, x0, HW, 0, true);
    if (T2b) fill_any<T, TAIL, CH, LB>(tw, T2b, st2_m, Mb, x0, HW, Mb, false);
#pragma unroll
    for (int k = 0; k < kWords; ++k) area_t += __builtin_popcount(tw.lo[k]) + __builtin_popcount(tw.hi[k]);
#pragma unroll
    for (int g = 0; g < NG; ++g) {     // no `break` here: the loop must unroll fully or acc[g][m] lands in scratch
        int nn = Nb - g * kWave;
        if (nn > kWave) nn = kWave;
        if (nn > 0) {
        BitTileT<CH> pw;
        fill_any<T, TAIL, CH, LB>(pw, Pb + (int64_t)g * kWave * sp_n, sp_n, nn, x0, HW, 0, true);
#pragma unroll
        for (int k = 0; k < kWords; ++k) area_p[g] += __builtin_popcount(pw.lo[k]) + __builtin_popcount(pw.hi[k]);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (m < Mrows) {
                unsigned a = acc[g][m];
#pragma unroll
                for (int k = 0; k < kWords; ++k) {
                    const int tl = __builtin_amdgcn_readlane(tw.lo[k], m);
                    const int th = __builtin_amdgcn_readlane(tw.hi[k], m);
                    a += __builtin_popcount(pw.lo[k] & tl) + __builtin_popcount(pw.hi[k] & th);
                }
                acc[g][m] = a;
            }
        }
        }
    }
}

// grid = (splits, B); block = 256.  inter / area_* must be zero on entry (the launcher memsets).
// Handles the tile [n0, n0 + 64*NG) x [m0, m0 + MT) of the (proposal, template) table.


template <typename T, int MT, int NG, int CH = kChunk, int LB = kLoadBytes>
__device__ __forceinline__ void iou_counts_body(
    const T *__restrict__ masks_p, const T *__restrict__ masks_t, const T *__restrict__ masks_t2, int N, int M, int HW,
    int64_t sp_b, int64_t sp_n, int64_t st_b, int64_t st_m, int64_t st2_b, int64_t st2_m,
    const int32_t *__restrict__ n_valid, const int32_t *__restrict__ m_valid, int32_t *__restrict__ inter,
    int32_t *__restrict__ area_p, int32_t *__restrict__ area_t, int32_t *__restrict__ inter2,
    int32_t *__restrict__ area_t2, int n0, int m0, int chunks_per_wg, int write_area_p, int write_area_t,
    int xcd_remap, int sub_count, int n_sub, int bx, int by, int gx, int gy) {
    int b, range;
    xcd_frame_range(xcd_remap, b, range, bx, by, gx, gy);
    // small batches: the proposal tile is cut into sub_count sub-tiles of n_sub planes, one workgroup each, so that a
    // handful of frames still spreads over the chip (the templates are re-read per sub-tile: latency, not bandwidth,
    // is what a B = 1 launch pays for)
    const int sub = range % sub_count;
    range /= sub_count;
    n0 += sub * n_sub;
    if (sub != 0) write_area_t = 0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int Nb = n_valid ? n_valid[b] : N;
    int Mb = m_valid ? m_valid[b] : M;
    Nb = min(min(Nb - n0, NG * kWave - sub * n_sub), n_sub);
    Mb = min(Mb - m0, masks_t2 ? MT / 2 : MT);
    if (Nb <= 0 || Mb <= 0) return;
    const int Mrows = masks_t2 ? 2 * Mb : Mb;                    // rows of the (template | target) tile
    const T *Pb = frame_base(masks_p, b, sp_b) + (int64_t)n0 * sp_n;
    const T *Tb = masks_t + (int64_t)b * st_b + (int64_t)m0 * st_m;
    const T *T2b = masks_t2 ? masks_t2 + (int64_t)b * st2_b + (int64_t)m0 * st2_m : nullptr;

    unsigned acc[NG][MT];
    unsigned ap[NG];
    unsigned at = 0;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        ap[g] = 0;
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[g][m] = 0;
    }

    const int full_chunks = HW / CH;
    const int nchunks = (HW + CH - 1) / CH;
    const int c_begin = range * chunks_per_wg;
    const int c_end = min(nchunks, c_begin + chunks_per_wg);
    for (int c = c_begin + wave; c < c_end; c += kCostThreads / kWave) {
        const int x0 = c * CH;
        if (c < full_chunks)
            process_chunk<T, MT, NG, false, CH, LB>(Pb, Tb, T2b, sp_n, st_m, st2_m, Nb, Mb, Mrows, x0, HW, acc, ap, at);
        else
            process_chunk<T, MT, NG, true, CH, LB>(Pb, Tb, T2b, sp_n, st_m, st2_m, Nb, Mb, Mrows, x0, HW, acc, ap, at);
    }

    // fold the 4 waves (integer LDS atomics), then one global atomic per entry
    __shared__ unsigned red[(MT + 1) * NG * kWave + kWave];
    unsigned *red_at = red + (MT + 1) * NG * kWave;
    for (int i = threadIdx.x; i < (MT + 1) * NG * kWave + kWave; i += kCostThreads) red[i] = 0;
    __syncthreads();
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int col = g * kWave + lane;
        if (col < Nb) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
                if (m < Mrows && acc[g][m]) atomicAdd(&red[m * NG * kWave + col], acc[g][m]);
            if (ap[g]) atomicAdd(&red[MT * NG * kWave + col], ap[g]);
        }
    }
    if (lane < Mrows && at) atomicAdd(&red_at[lane], at);
    __syncthreads();
    int32_t *inter_b = inter + (int64_t)b * M * N;
    int32_t *inter2_b = inter2 ? inter2 + (int64_t)b * M * N : nullptr;
    for (int i = threadIdx.x; i < MT * NG * kWave; i += kCostThreads) {
        const int m = i / (NG * kWave), col = i % (NG * kWave);
        if (col < Nb && red[i]) {
            if (m < Mb) atomicAdd(&inter_b[(int64_t)(m0 + m) * N + n0 + col], (int)red[i]);
            else if (m < Mrows) atomicAdd(&inter2_b[(int64_t)(m0 + m - Mb) * N + n0 + col], (int)red[i]);
        }
    }
    if (write_area_p)
        for (int col = threadIdx.x; col < Nb; col += kCostThreads)
            if (red[MT * NG * kWave + col]) atomicAdd(&area_p[(int64_t)b * N + n0 + col], (int)red[MT * NG * kWave + col]);
    if (write_area_t && threadIdx.x < Mrows && red_at[threadIdx.x]) {
        if (threadIdx.x < Mb) atomicAdd(&area_t[(int64_t)b * M + m0 + threadIdx.x], (int)red_at[threadIdx.x]);
        else atomicAdd(&area_t2[(int64_t)b * M + m0 + threadIdx.x - Mb], (int)red_at[threadIdx.x]);
    }
}

template <typename T, int MT, int NG, int CH = kChunk, int LB = kLoadBytes>
__global__ __launch_bounds__(kCostThreads) void iou_counts_kernel(
    const T *__restrict__ masks_p, const T *__restrict__ masks_t, const T *__restrict__ masks_t2, int N, int M, int HW,
    int64_t sp_b, int64_t sp_n, int64_t st_b, int64_t st_m, int64_t st2_b, int64_t st2_m,
    const int32_t *__restrict__ n_valid, const int32_t *__restrict__ m_valid, int32_t *__restrict__ inter,
    int32_t *__restrict__ area_p, int32_t *__restrict__ area_t, int32_t *__restrict__ inter2,
    int32_t *__restrict__ area_t2, int n0, int m0, int chunks_per_wg, int write_area_p, int write_area_t,
    int xcd_remap, int sub_count, int n_sub) {
    iou_counts_body<T, MT, NG, CH, LB>(masks_p, masks_t, masks_t2, N, M, HW, sp_b, sp_n, st_b, st_m, st2_b, st2_m, n_valid,
                                       m_valid, inter, area_p, area_t, inter2, area_t2, n0, m0, chunks_per_wg, write_area_p,
                                       write_area_t, xcd_remap, sub_count, n_sub, blockIdx.x, blockIdx.y, gridDim.x,
                                       gridDim.y);
}

// ---------------------------------------------------------------------------------------------
// Small-batch front kernel of dmm_match_forward (DMM_OPT_SMALL_FUSED): the feature similarity and the counts of a handful
// of frames in ONE launch.  At B = 1 the similarity is 2 workgroups busy for ~10 us (a dependent chain, 120 KB of
// features) and the counts ~450 workgroups for ~13 us (16 MB of planes); as two launches they queue behind each other.
// Here the first cos_parts workgroups of every frame run cosine_lanes_body, the rest iou_counts_body in its small-batch
// shape (one 16-byte lane load per plane and chunk, sub-tiles of proposals) -- neither depends on the other, both feed the
// solver.  The count tables must be zero on entry (dmm_front_small clears them with one launch in front).
// grid = (cos_parts + splits * sub_count, B), block = 256, dynamic LDS = lanes_geom<LPC>().lds.
// ---------------------------------------------------------------------------------------------
template <int LPC, typename T, int MT>
__global__ __launch_bounds__(kCostThreads) void front_small_kernel(
    const float *__restrict__ feat_t, const float *__restrict__ feat_p, float *__restrict__ cos_out, int cos_parts,
    int cos_waves, const T *__restrict__ masks_p, const T *__restrict__ masks_t, const T *__restrict__ masks_t2, int N, int M,
    int HW,
    int64_t sp_b, int64_t sp_n, int64_t st_b, int64_t st_m, int64_t st2_b, int64_t st2_m,
    const int32_t *__restrict__ n_valid, const int32_t *__restrict__ m_valid, int32_t *__restrict__ inter,
    int32_t *__restrict__ area_p, int32_t *__restrict__ area_t, int32_t *__restrict__ inter2,
    int32_t *__restrict__ area_t2, int n0, int m0, int chunks_per_wg, int write_area_p, int write_area_t,
    int xcd_remap, int sub_count, int n_sub) {
    extern __shared__ __attribute__((aligned(16))) float front_lds[];
    if ((int)blockIdx.x < cos_parts) {
        cosine_lanes_body<LPC>(feat_t, feat_p, N, M, cos_out, front_lds, blockIdx.x, cos_parts, blockIdx.y, cos_waves);
        return;
    }
    // every count argument arrives as a kernel argument, as in iou_counts_kernel (with the unused ones as literals the
    // optimizer of this toolchain crashed)
    constexpr int CHS = 64 * MaskIO<T>::kVec, LBS = 16384;
    iou_counts_body<T, MT, 1, CHS, LBS>(masks_p, masks_t, masks_t2, N, M, HW, sp_b, sp_n, st_b, st_m, st2_b, st2_m, n_valid,
                                        m_valid, inter, area_p, area_t, inter2, area_t2, n0, m0, chunks_per_wg, write_area_p,
                                        write_area_t, xcd_remap, sub_count, n_sub, (int)blockIdx.x - cos_parts, blockIdx.y,
                                        (int)gridDim.x - cos_parts, gridDim.y);
}

template <typename T, int MT>
static int launch_front_small(const float *feat_t, const float *feat_p, float *cos_out, const T *masks_p, const T *masks_t,
                              const T *masks_t2, int B, int N, int M, int HW, int64_t sp_b, int64_t sp_n, int64_t st_b,
                              int64_t st_m, int64_t st2_b, int64_t st2_m, int32_t *inter, int32_t *area_p, int32_t *area_t,
                              int32_t *inter2, int32_t *area_t2, bool tables_zero, hipStream_t stream) {
    constexpr int LPC = 32;                                          // D = 512, the model's ROI feature width
    // similarity workgroups: 2 of the 4 waves take steps (every workgroup of the launch carries their LDS; with 4 wave
    // buffers only 2 workgroups fit a CU, with 2 three do)
    typedef CfGeom<LPC> G;
    const int A = N >= 8 ? 32 * (N / 32) : 4 * (N / 4), S = (A + 7) / 8 + (N - A + 7) / 8;
    LanesGeom g;
    g.nw = S < 2 ? S : 2;
    g.parts = (S + g.nw - 1) / g.nw;
    g.lds = sizeof(float) * ((size_t)M * G::PITCH + 32 + (size_t)g.nw * G::WAVE_FLOATS);
    // the small-batch shape of launch_tile below
    constexpr int CHS = 64 * MaskIO<T>::kVec;
    const int small_wgs = opt(DMM_OPT_COST_SMALL_WGS);
    const int nch = (HW + CHS - 1) / CHS;
    const int splits_s = (nch + kCostThreads / kWave - 1) / (kCostThreads / kWave);
    int sub_count = 1, n_sub = kWave;
    if ((int64_t)B * splits_s < small_wgs && N > 8) {
        sub_count = (int)((small_wgs + (int64_t)B * splits_s - 1) / ((int64_t)B * splits_s));
        const int max_sub = (N + 7) / 8;
        if (sub_count > max_sub) sub_count = max_sub;
        n_sub = (N + sub_count - 1) / sub_count;
        sub_count = (N + n_sub - 1) / n_sub;
    }
    // every workgroup of the launch carries the similarity's LDS, so only a few fit a CU; the launch must stay ONE
    // resident wave of workgroups (a second wave would queue the counts behind the similarity's 10 us):
    // fewer sub-tiles of proposals if that does it, the separate launches otherwise
    {
        const int64_t per_cu = (int64_t)(160 * 1024) / (int64_t)(g.lds + 8 * 1024);
        const int64_t slots = 256 * (per_cu < 1 ? 1 : per_cu) - (int64_t)B * g.parts;
        while (sub_count > 1 && (int64_t)B * splits_s * sub_count > slots) {
            --sub_count;
            n_sub = (N + sub_count - 1) / sub_count;
            sub_count = (N + n_sub - 1) / n_sub;
        }
        if ((int64_t)B * splits_s * sub_count > slots) return DMM_ERR_UNSUPPORTED;
    }
    if (g.lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)front_small_kernel<LPC, T, MT>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);
        if (e != hipSuccess) { set_last_hip_error((int)e); return DMM_ERR_LAUNCH; }
    }
    // (training: the targets' tables inter2 | area_t2 follow the three tables in the caller's block -- one clearing launch)
    if (!tables_zero)
        DMM_HIP_TRY(zero_async(inter, sizeof(int32_t) * (((size_t)B * M * N + (size_t)B * M) * (masks_t2 ? 2 : 1) + (size_t)B * N),
                               stream));
    hipLaunchKernelGGL((front_small_kernel<LPC, T, MT>), dim3(g.parts + splits_s * sub_count, B), dim3(kCostThreads), g.lds,
                       stream, feat_t, feat_p, cos_out, g.parts, g.nw, masks_p, masks_t, masks_t2, N, M, HW, sp_b, sp_n,
                       st_b, st_m, st2_b, st2_m, (const int32_t *)nullptr, (const int32_t *)nullptr, inter, area_p,
                       area_t, inter2, area_t2, 0, 0, kCostThreads / kWave, 1, 1, 0, sub_count, n_sub);
    return check_launch();
}

// DMM_ERR_UNSUPPORTED (nothing launched) outside its envelope: B <= DMM_OPT_COST_TINY_FRAMES dense frames, N <= 64,
// M <= 16, D = 512, float / half / bfloat16 planes, the three tables contiguous (dmm_match_forward's workspace).
// tables_zero: the caller vouches that the tables are zero already (dmm_match_forward_ws) -- no clearing launch.
// masks_t2 (training, dmm_match_train_forward): a second template set -- the targets of compute_matching_loss -- rides in
// the same tile (rows M..2M-1), M <= 8; its tables inter2 | area_t2 must follow area_t in the same block.
int front_small_launch(const void *masks_p, const void *masks_t, const void *masks_t2, int dtype, const float *feat_t,
                       const float *feat_p, int B, int N, int M, int HW, int D, int64_t sp_b, int64_t sp_n, int64_t st_b,
                       int64_t st_m, int64_t st2_b, int64_t st2_m, float *cos_out, int32_t *inter, int32_t *area_p,
                       int32_t *area_t, int32_t *inter2, int32_t *area_t2, bool tables_zero, hipStream_t stream) {
    if (opt(DMM_OPT_SMALL_FUSED) != 1 || B > opt(DMM_OPT_COST_TINY_FRAMES) || opt(DMM_OPT_COST_KERNEL) == 1)
        return DMM_ERR_UNSUPPORTED;
    const int rows = masks_t2 ? 2 * M : M;
    if (D != 512 || N < 2 || N > 64 || M < 1 || rows > 16 || HW <= 0 || sp_n < HW || st_m < HW) return DMM_ERR_UNSUPPORTED;
    if (area_p != inter + (size_t)B * M * N || area_t != area_p + (size_t)B * N) return DMM_ERR_UNSUPPORTED;
    if (masks_t2 && (st2_m < HW || inter2 != area_t + (size_t)B * M || area_t2 != inter2 + (size_t)B * M * N))
        return DMM_ERR_UNSUPPORTED;
#define DMM_FRONT_CASE(T_)                                                                                              \
    return rows <= 8 ? launch_front_small<T_, 8>(feat_t, feat_p, cos_out, (const T_ *)masks_p, (const T_ *)masks_t,       \
                                                 (const T_ *)masks_t2, B, N, M, HW, sp_b, sp_n, st_b, st_m, st2_b, st2_m, \
                                                 inter, area_p, area_t, inter2, area_t2, tables_zero, stream)            \
                     : launch_front_small<T_, 16>(feat_t, feat_p, cos_out, (const T_ *)masks_p, (const T_ *)masks_t,      \
                                                  (const T_ *)masks_t2, B, N, M, HW, sp_b, sp_n, st_b, st_m, st2_b, st2_m, \
                                                  inter, area_p, area_t, inter2, area_t2, tables_zero, stream)
    switch (dtype) {
        case DMM_F32: DMM_FRONT_CASE(float);
        case DMM_F16: DMM_FRONT_CASE(f16_t);
        case DMM_BF16: DMM_FRONT_CASE(bf16_t);
        default: return DMM_ERR_UNSUPPORTED;
    }
#undef DMM_FRONT_CASE
}

template <typename T, int MT, int NG>
static int launch_tile(const T *masks_p, const T *masks_t, const T *masks_t2, int B, int N, int M, int HW, int64_t sp_b,
                       int64_t sp_n, int64_t st_b, int64_t st_m, int64_t st2_b, int64_t st2_m, const int32_t *n_valid,
                       const int32_t *m_valid, int32_t *inter, int32_t *area_p, int32_t *area_t, int32_t *inter2,
                       int32_t *area_t2, int n0, int m0, int wap, int wat, hipStream_t stream) {
    const int target_wgs = opt(DMM_OPT_COST_WGS);
    // workgroups below which a launch is split further (one chunk per workgroup, sub-tiles of proposals).  Every sub-tile
    // re-reads the frame's template planes, so the target must not be higher than it takes to fill the chip: measured per
    // call (cosine + counts + solver + mix) at B = 1 / 4 / 8 / 64 frames of 50 x 10, 255 x 255: 1024 -> 0.109 / 0.124 / 0.137
    // / 0.473 ms, 512 -> 0.109 / 0.117 / 0.134 / 0.345, 256 -> 0.106 / 0.121 / 0.135 / 0.346, 2048 -> 0.109 / 0.133 / 0.142 / 0.474
    const int small_wgs = opt(DMM_OPT_COST_SMALL_WGS);
    const int tiny_frames = opt(DMM_OPT_COST_TINY_FRAMES);       // default 8: B = 8 0.162 ms per sequence with it, 0.191 without
    const int xcd_remap = opt(DMM_OPT_COST_XCD);
    // A handful of frames (the product's B = 1 / B = 4 calls) is a LATENCY problem: with 1024-pixel chunks one frame
    // has 64 of them, one per workgroup, so three waves of every workgroup idled and the working one went through 9
    // dependent load batches (10 template planes + 8 proposals, 2 planes in flight): 19 us at B = 1.  The same kernel
    // with one 16-byte lane load per plane and chunk (256 / 512 pixels) and 16 planes in flight gives all four waves a
    // chunk and needs 2 batches: 11 us at B = 1, 24 us at B = 4 (profiles/r02_latency_B*_timeline.txt).
    if constexpr (NG == 1 && MT <= 16 && !std::is_same<T, packed_t>::value) {
        if (B <= tiny_frames) {
            constexpr int CHS = 64 * MaskIO<T>::kVec, LBS = 16384;
            const int nch = (HW + CHS - 1) / CHS;
            const int splits_s = (nch + kCostThreads / kWave - 1) / (kCostThreads / kWave);
            const int ntile_s = (N - n0) < kWave ? (N - n0) : kWave;
            int sub_count = 1, n_sub = kWave;
            if ((int64_t)B * splits_s < small_wgs && ntile_s > 8) {
                sub_count = (int)((small_wgs + (int64_t)B * splits_s - 1) / ((int64_t)B * splits_s));
                const int max_sub = (ntile_s + 7) / 8;
                if (sub_count > max_sub) sub_count = max_sub;
                n_sub = (ntile_s + sub_count - 1) / sub_count;
                sub_count = (ntile_s + n_sub - 1) / n_sub;
            }
            hipLaunchKernelGGL((iou_counts_kernel<T, MT, NG, CHS, LBS>), dim3(splits_s * sub_count, B), dim3(kCostThreads), 0,
                               stream, masks_p, masks_t, masks_t2, N, M, HW, sp_b, sp_n, st_b, st_m, st2_b, st2_m, n_valid,
                               m_valid, inter, area_p, area_t, inter2, area_t2, n0, m0, kCostThreads / kWave, wap, wat,
                               xcd_remap, sub_count, n_sub);
            return check_launch();
        }
    }
    const int nchunks = (HW + kChunk - 1) / kChunk;
    // ~8192 workgroups, at least 1 chunk per wave: small workgroups keep the tail of the launch short and measured
    // best (B = 1024: 5.8 / 6.0 / 6.3 / 6.6 / 6.4 TB/s at 1k / 2k / 4k / 8k / 16k workgroups)
    int splits = (target_wgs + B - 1) / B;
    int max_splits = (nchunks + kCostThreads / kWave - 1) / (kCostThreads / kWave);
    // a few frames only (the product's B = 1 / B = 4 calls): one chunk per workgroup (waves 1-3 of it idle) ...
    if ((int64_t)B * max_splits < small_wgs) max_splits = nchunks;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    const int chunks_per_wg = (nchunks + splits - 1) / splits;
    splits = (nchunks + chunks_per_wg - 1) / chunks_per_wg;
    // ... and sub-tiles of >= 8 proposals until ~small_wgs workgroups exist (B = 1, N = 50: 16 workgroups took 33 us,
    // 448 take 17; B = 4: 53 us with 64)
    const int ntile = (N - n0) < NG * kWave ? (N - n0) : NG * kWave;
    int sub_count = 1, n_sub = NG * kWave;
    if ((int64_t)B * splits < small_wgs && ntile > 8) {
        sub_count = (int)((small_wgs + (int64_t)B * splits - 1) / ((int64_t)B * splits));
        const int max_sub = (ntile + 7) / 8;
        if (sub_count > max_sub) sub_count = max_sub;
        n_sub = (ntile + sub_count - 1) / sub_count;
        sub_count = (ntile + n_sub - 1) / n_sub;
    }
    dim3 grid(splits * sub_count, B);
    hipLaunchKernelGGL((iou_counts_kernel<T, MT, NG>), grid, dim3(kCostThreads), 0, stream, masks_p, masks_t, masks_t2, N,
                       M, HW, sp_b, sp_n, st_b, st_m, st2_b, st2_m, n_valid, m_valid, inter, area_p, area_t, inter2,
                       area_t2, n0, m0, chunks_per_wg, wap, wat, xcd_remap, sub_count, n_sub);
    return check_launch();
}

// ---------------------------------------------------------------------------------------------
// Template-lane variant (large tables: many proposals and / or > 16 template rows).
// The kernel above parks BOTH bit tiles in registers (lane = proposal) and keeps MT x NG accumulators per lane; at
// N = 200, M = 20 that is 2 waves / SIMD, 128-proposal tiles (templates re-read per tile) and a VALU share that no
// longer hides under the loads.  Here only the TEMPLATE tile is parked (lane = template row, + one all-ones lane
// whose count is the proposal's area) and proposal planes are never parked at all: the ballot of a proposal word
// is a wave-uniform SGPR pair, v_and + v_bcnt against the parked template words give the counts of that proposal
// against every template row at once (4 VALU ops per word), and one LDS atomic per (proposal, chunk) folds the
// lane vector into the workgroup's [proposal][row] table.  No accumulator registers, no proposal tiling, no
// per-tile template re-reads; ~80 VGPRs.
// ---------------------------------------------------------------------------------------------
template <typename T, bool TAIL>
__device__ __forceinline__ void tl_chunk(const T *Pb, const T *Tb, const T *T2b, int64_t sp_n, int64_t st_m,
                                         int64_t st2_m, int Nb, int Mb, int Mrows, int x0, int HW, unsigned *red, int RS,
                                         unsigned &area_t) {
    constexpr int E = MaskIO<T>::kVec;
    constexpr int SUB = kChunk / (64 * E);
    constexpr int kUnroll = DMM_TL_LOAD_BYTES / (kChunk * (int)sizeof(T));
    const int lane = threadIdx.x & 63;
    BitTile tw;
    fill_tile<T, TAIL>(tw, Tb, st_m, Mb, x0, HW, 0, true);
    if (T2b) fill_tile<T, TAIL>(tw, T2b, st2_m, Mb, x0, HW, Mb, false);
#pragma unroll
    for (int k = 0; k < kWords; ++k) area_t += __builtin_popcount(tw.lo[k]) + __builtin_popcount(tw.hi[k]);
    if (lane == Mrows) {                       // the all-ones row: popc(P & ~0) = area of the proposal
#pragma unroll
        for (int k = 0; k < kWords; ++k) { tw.lo[k] = -1; tw.hi[k] = -1; }
    }
    const int x = x0 + lane * E;
    typedef typename MaskIO<T>::Raw Raw;
    // Software pipeline: the loads of the NEXT kUnroll planes are in flight while the current ones are thresholded
    // and counted (ping-pong register sets A / B, no copies); without it a wave issued its next loads only after ~400
    // VALU instructions on the previous group.
    auto load_group = [&](Raw (&v)[kUnroll][SUB], int p0) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int p = p0 + u < Nb ? p0 + u : Nb - 1;
#pragma unroll
            for (int j = 0; j < SUB; ++j) v[u][j] = load_pixels<T, TAIL>(Pb + (int64_t)p * sp_n, x + j * 64 * E, HW);
        }
    };
    auto count_group = [&](const Raw (&v)[kUnroll][SUB], int p0) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            if (p0 + u < Nb) {
                unsigned a0 = 0, a1 = 0;                     // two chains of accumulating v_bcnt (dst = popc(src) + acc)
#pragma unroll
                for (int j = 0; j < SUB; ++j)
#pragma unroll
                    for (int k = 0; k < E; ++k) {
                        const unsigned long long b = __ballot(MaskIO<T>::gt_half(v[u][j], k));
                        const int lo = tw.lo[E * j + k] & (int)(unsigned)b, hi = tw.hi[E * j + k] & (int)(unsigned)(b >> 32);
                        asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a0) : "v"(lo));
                        asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a1) : "v"(hi));
                    }
                if (lane <= Mrows) atomicAdd(&red[(p0 + u) * RS + lane], a0 + a1);
            }
        }
    };
    Raw va[kUnroll][SUB], vb[kUnroll][SUB];
    load_group(va, 0);
    for (int p0 = 0; p0 < Nb; p0 += 2 * kUnroll) {
        if (p0 + kUnroll < Nb) load_group(vb, p0 + kUnroll);
        count_group(va, p0);
        if (p0 + kUnroll >= Nb) break;
        if (p0 + 2 * kUnroll < Nb) load_group(va, p0 + 2 * kUnroll);
        count_group(vb, p0 + kUnroll);
    }
}

// grid = (splits, B); block = 256; dynamic LDS = (nt * RS + 64) * 4 bytes, RS = Mrows_max + 1.
template <typename T>
__global__ __launch_bounds__(kCostThreads, DMM_TL_MIN_WAVES) void iou_counts_tl_kernel(
    const T *__restrict__ masks_p, const T *__restrict__ masks_t, const T *__restrict__ masks_t2, int N, int M, int HW,
    int64_t sp_b, int64_t sp_n, int64_t st_b, int64_t st_m, int64_t st2_b, int64_t st2_m,
    const int32_t *__restrict__ n_valid, const int32_t *__restrict__ m_valid, int32_t *__restrict__ inter,
    int32_t *__restrict__ area_p, int32_t *__restrict__ area_t, int32_t *__restrict__ inter2,
    int32_t *__restrict__ area_t2, int n0, int m0, int nt, int mt, int chunks_per_wg, int write_area_p,
    int write_area_t, int RS, int xcd_remap) {
    extern __shared__ unsigned tl_red[];
    int b, range;
    xcd_frame_range(xcd_remap, b, range);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int Nb = n_valid ? n_valid[b] : N;
    int Mb = m_valid ? m_valid[b] : M;
    Nb = min(Nb - n0, nt);
    Mb = min(Mb - m0, mt);
    if (Nb <= 0 || Mb <= 0) return;
    const int Mrows = masks_t2 ? 2 * Mb : Mb;
    const T *Pb = frame_base(masks_p, b, sp_b) + (int64_t)n0 * sp_n;
    const T *Tb = masks_t + (int64_t)b * st_b + (int64_t)m0 * st_m;
    const T *T2b = masks_t2 ? masks_t2 + (int64_t)b * st2_b + (int64_t)m0 * st2_m : nullptr;
    unsigned *red_at = tl_red + nt * RS;
    for (int i = threadIdx.x; i < nt * RS + kWave; i += kCostThreads) tl_red[i] = 0;
    __syncthreads();
    unsigned at = 0;
    const int full_chunks = HW / kChunk;
    const int nchunks = (HW + kChunk - 1) / kChunk;
    const int c_begin = range * chunks_per_wg;
    const int c_end = min(nchunks, c_begin + chunks_per_wg);
    for (int c = c_begin + wave; c < c_end; c += kCostThreads / kWave) {
        const int x0 = c * kChunk;
        if (c < full_chunks) tl_chunk<T, false>(Pb, Tb, T2b, sp_n, st_m, st2_m, Nb, Mb, Mrows, x0, HW, tl_red, RS, at);
        else tl_chunk<T, true>(Pb, Tb, T2b, sp_n, st_m, st2_m, Nb, Mb, Mrows, x0, HW, tl_red, RS, at);
    }
    if (lane < Mrows && at) atomicAdd(&red_at[lane], at);
    __syncthreads();
    int32_t *inter_b = inter + (int64_t)b * M * N;
    int32_t *inter2_b = inter2 ? inter2 + (int64_t)b * M * N : nullptr;
    // row-major over the OUTPUT tables (consecutive threads = consecutive proposals of one row): a wave's 64 atomics
    // then fall into 2-3 lines instead of 64 (the [proposal][row] order of tl_red made every one its own L2
    // transaction: 4200 per workgroup at config 5, and the launch slowed with every extra workgroup);
    // RS is odd or coprime with the 64 banks for the shipped shapes, the strided LDS reads stay cheap
    for (int i = threadIdx.x; i < (Mrows + 1) * Nb; i += kCostThreads) {
        const int r = i / Nb, p = i - r * Nb;
        const unsigned v = tl_red[p * RS + r];
        if (!v) continue;
        if (r < Mb) atomicAdd(&inter_b[(int64_t)(m0 + r) * N + n0 + p], (int)v);
        else if (r < Mrows) atomicAdd(&inter2_b[(int64_t)(m0 + r - Mb) * N + n0 + p], (int)v);
        else if (write_area_p) atomicAdd(&area_p[(int64_t)b * N + n0 + p], (int)v);
    }
    if (write_area_t && threadIdx.x < Mrows && red_at[threadIdx.x]) {
        if (threadIdx.x < Mb) atomicAdd(&area_t[(int64_t)b * M + m0 + threadIdx.x], (int)red_at[threadIdx.x]);
        else atomicAdd(&area_t2[(int64_t)b * M + m0 + threadIdx.x - Mb], (int)red_at[threadIdx.x]);
    }
}

template <typename T>
static int launch_tl(const T *masks_p, const T *masks_t, const T *masks_t2, int B, int N, int M, int HW, int64_t sp_b,
                     int64_t sp_n, int64_t st_b, int64_t st_m, int64_t st2_b, int64_t st2_m, const int32_t *n_valid,
                     const int32_t *m_valid, int32_t *inter, int32_t *area_p, int32_t *area_t, int32_t *inter2,
                     int32_t *area_t2, int n0, int m0, int nt, int mt, int wap, int wat, hipStream_t stream) {
    const int nchunks = (HW + kChunk - 1) / kChunk;
    // FEW, long-lived workgroups: every one zeroes and flushes its own [proposal][row] table (4200 entries at config 5),
    // and this kernel's rate does not follow its occupancy (2 waves per SIMD stream as fast as 4).  Measured at config 5,
    // ms per launch at 512 / 2048 / 8192 workgroups: 512 frames 2.42 / 2.48 / 2.53, 128 frames 0.68 / 0.71 / 0.71.
    const int target_wgs = opt(DMM_OPT_COST_TL_WGS);
    int splits = (target_wgs + B - 1) / B;
    const int max_splits = (nchunks + kCostThreads / kWave - 1) / (kCostThreads / kWave);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    const int chunks_per_wg = (nchunks + splits - 1) / splits;
    splits = (nchunks + chunks_per_wg - 1) / chunks_per_wg;
    const int xcd_remap = opt(DMM_OPT_COST_XCD);
    const int RS = (masks_t2 ? 2 * mt : mt) + 1;
    const size_t lds = sizeof(unsigned) * ((size_t)nt * RS + kWave);
    hipLaunchKernelGGL((iou_counts_tl_kernel<T>), dim3(splits, B), dim3(kCostThreads), lds, stream, masks_p, masks_t,
                       masks_t2, N, M, HW, sp_b, sp_n, st_b, st_m, st2_b, st2_m, n_valid, m_valid, inter, area_p, area_t,
                       inter2, area_t2, n0, m0, nt, mt, chunks_per_wg, wap, wat, RS, xcd_remap);
    return check_launch();
}

// set (per thread, for the duration of one call) by iou_counts_prezeroed
static thread_local bool g_tables_prezeroed = false;

template <typename T>
static int iou_counts_typed(const T *masks_p, const T *masks_t, const T *masks_t2, int B, int N, int M, int HW,
                            int64_t sp_b, int64_t sp_n, int64_t st_b, int64_t st_m, int64_t st2_b, int64_t st2_m,
                            const int32_t *n_valid, const int32_t *m_valid, int32_t *inter, int32_t *area_p,
                            int32_t *area_t, int32_t *inter2, int32_t *area_t2, hipStream_t stream) {
    if (g_tables_prezeroed) {
        // dmm_match_forward / dmm_match_train_forward: the feature-similarity launch in front of this one already cleared
        // the tables (all five of them in the dual form)
    } else if (area_p == inter + (size_t)B * M * N && area_t == area_p + (size_t)B * N) {
        // the three tables are one contiguous block (dmm_match_forward's workspace): one memset node
        DMM_HIP_TRY(zero_async(inter, sizeof(int32_t) * ((size_t)B * M * N + (size_t)B * N + (size_t)B * M), stream));
    } else {
        DMM_HIP_TRY(zero_async(inter, sizeof(int32_t) * (size_t)B * M * N, stream));
        DMM_HIP_TRY(zero_async(area_p, sizeof(int32_t) * (size_t)B * N, stream));
        DMM_HIP_TRY(zero_async(area_t, sizeof(int32_t) * (size_t)B * M, stream));
    }
    if (masks_t2 && !g_tables_prezeroed) {
        DMM_HIP_TRY(zero_async(inter2, sizeof(int32_t) * (size_t)B * M * N, stream));
        DMM_HIP_TRY(zero_async(area_t2, sizeof(int32_t) * (size_t)B * M, stream));
    }
    if (HW == 0) return DMM_OK;
    // Tile the (N, M) table over the compiled envelopes: <= 32 template rows per launch (<= 16 when a second
    // template set rides along: both sets share the tile) x <= 256 proposals.  Register budget decides the
    // proposal tile: accumulators are MT x NG per lane, and at MT > 16 a 4-group tile drops to 1 wave/SIMD
    // (measured 2.6 TB/s on config 5), so those shapes run as 128-proposal tiles (templates re-read once
    // per tile: +9 % bytes at N=200, M=20).
    // kernel choice: DMM_OPT_COST_KERNEL = 0 (register tiles) / 1 (template lanes) / -1 = by shape (the IoU tests pin both)
    const int kernel_mode = opt(DMM_OPT_COST_KERNEL);
    if constexpr (!std::is_same<T, packed_t>::value) {
        const int rows = masks_t2 ? 2 * M : M;
        const bool use_tl = kernel_mode == 1 || (kernel_mode < 0 && (rows > 16 || N > 128));
        if (use_tl) {
            const int mstep_tl = masks_t2 ? 16 : 32;
            for (int m0 = 0; m0 < M; m0 += mstep_tl) {
                const int mt = M - m0 < mstep_tl ? M - m0 : mstep_tl;
                for (int n0 = 0; n0 < N; n0 += 256) {
                    const int nt = N - n0 < 256 ? N - n0 : 256;
                    const int rc = launch_tl<T>(masks_p, masks_t, masks_t2, B, N, M, HW, sp_b, sp_n, st_b, st_m, st2_b,
                                                st2_m, n_valid, m_valid, inter, area_p, area_t, inter2, area_t2, n0, m0,
                                                nt, mt, m0 == 0, n0 == 0, stream);
                    if (rc != DMM_OK) return rc;
                }
            }
            return DMM_OK;
        }
    }
    const int mstep = masks_t2 ? 16 : 32;
    for (int m0 = 0; m0 < M; m0 += mstep) {
        const int mt = (M - m0 < mstep ? M - m0 : mstep) * (masks_t2 ? 2 : 1);
        const int nstep = mt > 16 ? 128 : 256;
        for (int n0 = 0; n0 < N; n0 += nstep) {
            const int nt = N - n0 < nstep ? N - n0 : nstep;
            const int wap = (m0 == 0), wat = (n0 == 0);
            int rc;
#define DMM_COST_CASE(MT_, NG_)                                                                                       \
    rc = launch_tile<T, MT_, NG_>(masks_p, masks_t, masks_t2, B, N, M, HW, sp_b, sp_n, st_b, st_m, st2_b, st2_m, n_valid, \
                                  m_valid, inter, area_p, area_t, inter2, area_t2, n0, m0, wap, wat, stream)
            if (nt <= 64) {
                if (mt <= 8) DMM_COST_CASE(8, 1);
                else if (mt <= 16) DMM_COST_CASE(16, 1);
                else if (mt <= 24) DMM_COST_CASE(24, 1);
                else DMM_COST_CASE(32, 1);
            } else if (nt <= 128) {
                if (mt <= 8) DMM_COST_CASE(8, 2);
                else if (mt <= 16) DMM_COST_CASE(16, 2);
                else if (mt <= 24) DMM_COST_CASE(24, 2);
                else DMM_COST_CASE(32, 2);
            } else {
                if (mt <= 8) DMM_COST_CASE(8, 4);
                else DMM_COST_CASE(16, 4);
            }
#undef DMM_COST_CASE
            if (rc != DMM_OK) return rc;
        }
    }
    return DMM_OK;
}

}  // namespace dmm

static int iou_counts_dispatch(const void *masks_p, const void *masks_t, const void *masks_t2, int dtype, int B, int N,
                               int M, int HW, int64_t sp_b, int64_t sp_n, int64_t st_b, int64_t st_m, int64_t st2_b,
                               int64_t st2_m, const int32_t *n_valid, const int32_t *m_valid, int32_t *inter,
                               int32_t *area_p, int32_t *area_t, int32_t *inter2, int32_t *area_t2,
                               dmm_stream_t stream) {
    if (B < 0 || N < 0 || M < 0 || HW < 0) return DMM_ERR_BAD_ARG;
    if (B == 0 || N == 0 || M == 0) return DMM_OK;
    if (!masks_p || !masks_t || !inter || !area_p || !area_t) return DMM_ERR_BAD_ARG;
    if (masks_t2 && (!inter2 || !area_t2)) return DMM_ERR_BAD_ARG;
    const int64_t min_stride = dtype == DMM_PACKED1 ? 4 * ((int64_t)(HW + 255) / 256) : HW;
    if (sp_n < min_stride || st_m < min_stride || (masks_t2 && st2_m < min_stride)) return DMM_ERR_BAD_ARG;
    if (B > 65535) {   // grid.y limit: run in batch slices
        const size_t es = dtype == DMM_F32 ? 4 : (dtype == DMM_PACKED1 ? 8 : 2);
        for (int b0 = 0; b0 < B; b0 += 65535) {
            const int nb = B - b0 < 65535 ? B - b0 : 65535;
            const int rc = iou_counts_dispatch(
                sp_b == dmm::kFrameTable ? (const char *)masks_p + sizeof(void *) * (size_t)b0
                                         : (const char *)masks_p + es * (size_t)b0 * sp_b,
                (const char *)masks_t + es * (size_t)b0 * st_b,
                masks_t2 ? (const char *)masks_t2 + es * (size_t)b0 * st2_b : nullptr, dtype, nb, N, M, HW, sp_b, sp_n,
                st_b, st_m, st2_b, st2_m, n_valid ? n_valid + b0 : nullptr, m_valid ? m_valid + b0 : nullptr,
                inter + (size_t)b0 * M * N, area_p + (size_t)b0 * N, area_t + (size_t)b0 * M,
                inter2 ? inter2 + (size_t)b0 * M * N : nullptr, area_t2 ? area_t2 + (size_t)b0 * M : nullptr, stream);
            if (rc != DMM_OK) return rc;
        }
        return DMM_OK;
    }
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case DMM_PACKED1:
            return dmm::iou_counts_typed<dmm::packed_t>((const dmm::packed_t *)masks_p, (const dmm::packed_t *)masks_t,
                                                        (const dmm::packed_t *)masks_t2, B, N, M, HW, sp_b, sp_n, st_b,
                                                        st_m, st2_b, st2_m, n_valid, m_valid, inter, area_p, area_t,
                                                        inter2, area_t2, s);
        case DMM_F32:
            return dmm::iou_counts_typed<float>((const float *)masks_p, (const float *)masks_t, (const float *)masks_t2, B,
                                                N, M, HW, sp_b, sp_n, st_b, st_m, st2_b, st2_m, n_valid, m_valid, inter,
                                                area_p, area_t, inter2, area_t2, s);
        case DMM_F16:
            return dmm::iou_counts_typed<dmm::f16_t>((const dmm::f16_t *)masks_p, (const dmm::f16_t *)masks_t,
                                                     (const dmm::f16_t *)masks_t2, B, N, M, HW, sp_b, sp_n, st_b, st_m,
                                                     st2_b, st2_m, n_valid, m_valid, inter, area_p, area_t, inter2,
                                                     area_t2, s);
        case DMM_BF16:
            return dmm::iou_counts_typed<dmm::bf16_t>((const dmm::bf16_t *)masks_p, (const dmm::bf16_t *)masks_t,
                                                      (const dmm::bf16_t *)masks_t2, B, N, M, HW, sp_b, sp_n, st_b, st_m,
                                                      st2_b, st2_m, n_valid, m_valid, inter, area_p, area_t, inter2,
                                                      area_t2, s);
        default:
            return DMM_ERR_BAD_ARG;
    }
}

extern "C" int dmm_iou_counts(const void *masks_p, const void *masks_t, int dtype, int B, int N, int M, int HW,
                              int64_t sp_b, int64_t sp_n, int64_t st_b, int64_t st_m, const int32_t *n_valid,
                              const int32_t *m_valid, int32_t *inter, int32_t *area_p, int32_t *area_t,
                              dmm_stream_t stream) {
    return iou_counts_dispatch(masks_p, masks_t, nullptr, dtype, B, N, M, HW, sp_b, sp_n, st_b, st_m, 0, 0, n_valid,
                               m_valid, inter, area_p, area_t, nullptr, nullptr, stream);
}

// dmm_iou_counts for a caller that has ALREADY zeroed inter / area_p / area_t on this stream (dmm_match_forward: the
// feature-similarity kernel clears them, one memset node less in front of the solver's dependent chain)
namespace dmm {
int iou_counts_prezeroed(const void *masks_p, const void *masks_t, int dtype, int B, int N, int M, int HW, int64_t sp_b,
                         int64_t sp_n, int64_t st_b, int64_t st_m, const int32_t *n_valid, const int32_t *m_valid,
                         int32_t *inter, int32_t *area_p, int32_t *area_t, dmm_stream_t stream) {
    g_tables_prezeroed = true;
    const int rc = dmm_iou_counts(masks_p, masks_t, dtype, B, N, M, HW, sp_b, sp_n, st_b, st_m, n_valid, m_valid, inter,
                                  area_p, area_t, stream);
    g_tables_prezeroed = false;
    return rc;
}
// the dual form (templates + targets) on tables the caller has already zeroed; sp_b may be kFrameTable (masks_p = the
// device table of per-frame base pointers)
int iou_counts_dual_prezeroed(const void *masks_p, const void *masks_t, const void *masks_t2, int dtype, int B, int N, int M,
                              int HW, int64_t sp_b, int64_t sp_n, int64_t st_b, int64_t st_m, int64_t st2_b, int64_t st2_m,
                              const int32_t *n_valid, const int32_t *m_valid, int32_t *inter, int32_t *area_p,
                              int32_t *area_t, int32_t *inter2, int32_t *area_t2, dmm_stream_t stream) {
    g_tables_prezeroed = true;
    const int rc = iou_counts_dispatch(masks_p, masks_t, masks_t2, dtype, B, N, M, HW, sp_b, sp_n, st_b, st_m, st2_b, st2_m,
                                       n_valid, m_valid, inter, area_p, area_t, inter2, area_t2, stream);
    g_tables_prezeroed = false;
    return rc;
}
}  // namespace dmm

extern "C" int dmm_iou_counts_dual(const void *masks_p, const void *masks_t, const void *masks_t2, int dtype, int B,
                                   int N, int M, int HW, int64_t sp_b, int64_t sp_n, int64_t st_b, int64_t st_m,
                                   int64_t st2_b, int64_t st2_m, const int32_t *n_valid, const int32_t *m_valid,
                                   int32_t *inter, int32_t *area_p, int32_t *area_t, int32_t *inter2, int32_t *area_t2,
                                   dmm_stream_t stream) {
    if (!masks_t2) return DMM_ERR_BAD_ARG;
    return iou_counts_dispatch(masks_p, masks_t, masks_t2, dtype, B, N, M, HW, sp_b, sp_n, st_b, st_m, st2_b, st2_m,
                               n_valid, m_valid, inter, area_p, area_t, inter2, area_t2, stream);
}

// ---- per-frame pointer tables for the proposal planes (the per-video tensors of DMM_Model: no batch copy) ----
extern "C" int dmm_iou_counts_frames(const void *const *masks_p_frames, const void *masks_t, int dtype, int B, int N,
                                     int M, int HW, int64_t sp_n, int64_t st_b, int64_t st_m, const int32_t *n_valid,
                                     const int32_t *m_valid, int32_t *inter, int32_t *area_p, int32_t *area_t,
                                     dmm_stream_t stream) {
    return iou_counts_dispatch((const void *)masks_p_frames, masks_t, nullptr, dtype, B, N, M, HW, dmm::kFrameTable, sp_n,
                               st_b, st_m, 0, 0, n_valid, m_valid, inter, area_p, area_t, nullptr, nullptr, stream);
}

extern "C" int dmm_iou_counts_dual_frames(const void *const *masks_p_frames, const void *masks_t, const void *masks_t2,
                                          int dtype, int B, int N, int M, int HW, int64_t sp_n, int64_t st_b,
                                          int64_t st_m, int64_t st2_b, int64_t st2_m, const int32_t *n_valid,
                                          const int32_t *m_valid, int32_t *inter, int32_t *area_p, int32_t *area_t,
                                          int32_t *inter2, int32_t *area_t2, dmm_stream_t stream) {
    if (!masks_t2) return DMM_ERR_BAD_ARG;
    return iou_counts_dispatch((const void *)masks_p_frames, masks_t, masks_t2, dtype, B, N, M, HW, dmm::kFrameTable,
                               sp_n, st_b, st_m, st2_b, st2_m, n_valid, m_valid, inter, area_p, area_t, inter2, area_t2,
                               stream);
}
