// dmm_common.h -- device helpers shared by the gfx950 kernels of libdmm_match.so.
// CDNA4 only: 64-lane wavefronts, GFX9 DPP controls (row_bcast15/31), no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dmm_match.h"

namespace dmm {
// Dispatch options (include/dmm_match.h (0), dmm_set_option): process-wide integers set THROUGH THE ABI -- the library
// never reads the environment.  One relaxed atomic load per use.
int opt(int key);


constexpr int kWave = 64;

// Frame addressing of the proposal planes.  Either the B frames are equally spaced (frame b at base + b * stride_b
// elements), or -- the per-video tensors of DMM_Model, the *_frames entry points of the C ABI -- `base` is a DEVICE
// ARRAY of B pointers to each frame's first plane; the launchers mark that with the stride sentinel kFrameTable.
constexpr int64_t kFrameTable = INT64_MIN;
template <typename T>
__device__ __forceinline__ const T *frame_base(const T *base, int b, int64_t stride_b) {
    // (the table is read AS an array of global-address-space pointers: the value then carries its address space into
    // the IR and the plane loads stay global_load; read as generic pointers they all became flat_load)
    typedef const __attribute__((address_space(1))) T *gptr_t;
    if (stride_b == kFrameTable) return (const T *)(reinterpret_cast<const gptr_t *>(base)[b]);
    return base + (int64_t)b * stride_b;
}

// 16-byte vectors that are only guaranteed 4-byte aligned: a 255x255 fp32 plane is 260100 B,
// so odd planes start 4/8/12 B off a 16-B boundary.  gfx950 global_load/store_dwordx4 only
// need dword alignment; the aligned(4) typedef makes hipcc emit them instead of 4 dword ops.
typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float float4a __attribute__((ext_vector_type(4)));   // naturally (16-byte) aligned
typedef uint32_t uint2u __attribute__((ext_vector_type(2), aligned(2)));  // 4 x 16-bit, 2-B aligned
typedef _Float16 half4u __attribute__((ext_vector_type(4), aligned(2)));
// 16-bit storage tags for mask planes (IEEE half / bfloat16); arithmetic is always fp32
struct f16_t { _Float16 v; };
struct bf16_t { uint16_t v; };
// DMM_PACKED1 storage: 1 bit per pixel, already thresholded (x > 0.5).  Layout ("ballot layout"): pixels are taken
// in blocks of 256; block q owns words 4q..4q+3; bit l of word 4q+k is pixel 256q + 4l + k (exactly what four wave
// ballots over a 16-byte-per-lane load produce).  A plane of HW pixels has 4*ceil(HW/256) words, pad bits are 0.
struct packed_t { unsigned long long v; };

// ---- DPP cross-lane primitives -------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_i32(int v, int old) {
    return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xF, false);
}
constexpr int DPP_XOR1 = 0xB1;         // quad_perm:[1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;         // quad_perm:[2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141; // row_half_mirror
constexpr int DPP_MIRROR = 0x140;      // row_mirror
constexpr int DPP_BCAST15 = 0x142;     // row_bcast:15
constexpr int DPP_BCAST31 = 0x143;     // row_bcast:31

__device__ __forceinline__ float readlane_f32(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// Sum over the 64 lanes, fixed tree order, result broadcast (wave-uniform).
// ((l, l^1), (.., ^2), half-mirror, mirror) inside each 16-lane row, then rows 0+1, 2+3, (01)+(23).
__device__ __forceinline__ float wave_sum(float v) {
    v = v + dpp_f32<DPP_XOR1>(v);
    v = v + dpp_f32<DPP_XOR2>(v);
    v = v + dpp_f32<DPP_HALF_MIRROR>(v);
    v = v + dpp_f32<DPP_MIRROR>(v);
    v = v + dpp_f32<DPP_BCAST15, 0xA>(v);
    v = v + dpp_f32<DPP_BCAST31, 0xC>(v);
    return readlane_f32(v, 63);
}

__device__ __forceinline__ float wave_max(float v) {
    // max needs the identity in disabled rows: use `old = v` so a disabled row keeps its value.
    auto step = [](float x, float y) { return x > y ? x : y; };
    v = step(v, dpp_f32<DPP_XOR1>(v));
    v = step(v, dpp_f32<DPP_XOR2>(v));
    v = step(v, dpp_f32<DPP_HALF_MIRROR>(v));
    v = step(v, dpp_f32<DPP_MIRROR>(v));
    float t = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), DPP_BCAST15, 0xA, 0xF, false));
    v = step(v, t);
    t = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), DPP_BCAST31, 0xC, 0xF, false));
    v = step(v, t);
    return readlane_f32(v, 63);
}

__device__ __forceinline__ float wave_min(float v) {
    auto step = [](float x, float y) { return x < y ? x : y; };
    v = step(v, dpp_f32<DPP_XOR1>(v));
    v = step(v, dpp_f32<DPP_XOR2>(v));
    v = step(v, dpp_f32<DPP_HALF_MIRROR>(v));
    v = step(v, dpp_f32<DPP_MIRROR>(v));
    float t = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), DPP_BCAST15, 0xA, 0xF, false));
    v = step(v, t);
    t = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), DPP_BCAST31, 0xC, 0xF, false));
    v = step(v, t);
    return readlane_f32(v, 63);
}

__device__ __forceinline__ int wave_min_i32(int v) {
    auto step = [](int x, int y) { return x < y ? x : y; };
    v = step(v, dpp_i32<DPP_XOR1>(v, v));
    v = step(v, dpp_i32<DPP_XOR2>(v, v));
    v = step(v, dpp_i32<DPP_HALF_MIRROR>(v, v));
    v = step(v, dpp_i32<DPP_MIRROR>(v, v));
    v = step(v, dpp_i32<DPP_BCAST15, 0xA>(v, v));
    v = step(v, dpp_i32<DPP_BCAST31, 0xC>(v, v));
    return __builtin_amdgcn_readlane(v, 63);
}

// ---- row-batched reductions: CNT independent values reduced with their DPP trees INTERLEAVED
// (step-major), so the 2-wait-state DPP hazards and the add latency of one tree are filled by the
// others.  Results are wave-uniform.  Same tree order as wave_sum / wave_max / wave_min.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_keep_f32(float v) {   // disabled rows keep their own value
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ float dpp_full_f32(float v) {   // full row mask, every source lane valid
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CNT>
__device__ __forceinline__ void wave_sum_rows(float (&v)[CNT]) {
#pragma unroll
    for (int i = 0; i < CNT; ++i) v[i] = v[i] + dpp_full_f32<DPP_XOR1>(v[i]);
#pragma unroll
    for (int i = 0; i < CNT; ++i) v[i] = v[i] + dpp_full_f32<DPP_XOR2>(v[i]);
#pragma unroll
    for (int i = 0; i < CNT; ++i) v[i] = v[i] + dpp_full_f32<DPP_HALF_MIRROR>(v[i]);
#pragma unroll
    for (int i = 0; i < CNT; ++i) v[i] = v[i] + dpp_full_f32<DPP_MIRROR>(v[i]);
#pragma unroll
    for (int i = 0; i < CNT; ++i) v[i] = v[i] + dpp_f32<DPP_BCAST15, 0xA>(v[i]);
#pragma unroll
    for (int i = 0; i < CNT; ++i) v[i] = v[i] + dpp_f32<DPP_BCAST31, 0xC>(v[i]);
#pragma unroll
    for (int i = 0; i < CNT; ++i) v[i] = readlane_f32(v[i], 63);
}
template <int CNT, typename OP>
__device__ __forceinline__ void wave_fold_rows(float (&v)[CNT], OP op) {
#pragma unroll
    for (int i = 0; i < CNT; ++i) v[i] = op(v[i], dpp_full_f32<DPP_XOR1>(v[i]));
#pragma unroll
    for (int i = 0; i < CNT; ++i) v[i] = op(v[i], dpp_full_f32<DPP_XOR2>(v[i]));
#pragma unroll
    for (int i = 0; i < CNT; ++i) v[i] = op(v[i], dpp_full_f32<DPP_HALF_MIRROR>(v[i]));
#pragma unroll
    for (int i = 0; i < CNT; ++i) v[i] = op(v[i], dpp_full_f32<DPP_MIRROR>(v[i]));
#pragma unroll
    for (int i = 0; i < CNT; ++i) v[i] = op(v[i], dpp_keep_f32<DPP_BCAST15, 0xA>(v[i]));
#pragma unroll
    for (int i = 0; i < CNT; ++i) v[i] = op(v[i], dpp_keep_f32<DPP_BCAST31, 0xC>(v[i]));
#pragma unroll
    for (int i = 0; i < CNT; ++i) v[i] = readlane_f32(v[i], 63);
}
struct op_fmax { __device__ __forceinline__ float operator()(float a, float b) const { return a > b ? a : b; } };
struct op_fmin { __device__ __forceinline__ float operator()(float a, float b) const { return a < b ? a : b; } };
template <int CNT> __device__ __forceinline__ void wave_max_rows(float (&v)[CNT]) { wave_fold_rows<CNT>(v, op_fmax()); }
template <int CNT> __device__ __forceinline__ void wave_min_rows(float (&v)[CNT]) { wave_fold_rows<CNT>(v, op_fmin()); }
template <int CNT>
__device__ __forceinline__ void wave_min_rows_i32(int (&v)[CNT]) {
#pragma unroll
    for (int i = 0; i < CNT; ++i) v[i] = wave_min_i32(v[i]);
}

// a / b for a loop-invariant b with rcp = RN(1/b): q0 = a*rcp, r = fma(-q0, b, a), q = fma(r, rcp, q0).
// Markstein's correction: the result equals the IEEE-754 correctly rounded a / b (what the reference's
// `/ X.shape[k]` computes).  Checked exhaustively enough for b = 1..256 over 5e8 operands by the
// oracle's dmmo_check_div_by_const (tests/test_host_logic.py); a, b are never denormal here.
__device__ __forceinline__ float div_by_const(float a, float b, float rcp) {
    const float q0 = a * rcp;
    const float r = __builtin_fmaf(-q0, b, a);
    return __builtin_fmaf(r, rcp, q0);
}

// Plane loads are NON-TEMPORAL: every mask plane is streamed once per kernel, and on gfx950 the `nt` load path
// sustains 7.1-7.2 TB/s where cached loads stop at 6.3 TB/s (tools/hbm_probe.py).
// ---- mask element loads: 4 consecutive pixels as fp32 (load4), or one full 16-byte lane load (loadv: 4 fp32 or
// 8 half/bfloat16 pixels) for the streaming kernels -------------------------------------------------------
// (-DDMM_PLANE_LOADS_CACHED builds the plane loads WITHOUT the non-temporal hint: the experiment of tools/mall_probe.py --
// do planes the cost pass streamed stay in the Infinity Cache for the mix?  The product keeps the hint.)
#ifdef DMM_PLANE_LOADS_CACHED
#define __builtin_nontemporal_load(p) (*(p))
#endif
typedef _Float16 half8u __attribute__((ext_vector_type(8), aligned(2)));
typedef uint32_t uint4u __attribute__((ext_vector_type(4), aligned(2)));
template <typename T> struct MaskIO;
template <> struct MaskIO<float> {
    static constexpr int kVec = 4;
    typedef float4u Raw;                                   // one 16-byte lane load, kept raw until it is consumed
    static __device__ __forceinline__ Raw load_raw(const float *p) { return __builtin_nontemporal_load(reinterpret_cast<const Raw *>(p)); }
    static __device__ __forceinline__ float elem(const Raw &r, int k) { return r[k]; }
    static __device__ __forceinline__ bool gt_half(const Raw &r, int k) { return r[k] > 0.5f; }   // x > 0.5 (strict)
    static __device__ __forceinline__ void loadv(const float *p, float (&v)[4]) { load4(p, v); }
    template <bool NT = true> static __device__ __forceinline__ void load4(const float *p, float (&v)[4]) {
        const float4u *q = reinterpret_cast<const float4u *>(p);
        float4u t;
        if (NT) t = __builtin_nontemporal_load(q);
        else t = *q;
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ float load1(const float *p) { return *p; }
};
template <> struct MaskIO<f16_t> {
    static constexpr int kVec = 8;
    typedef half8u Raw;
    static __device__ __forceinline__ Raw load_raw(const f16_t *p) { return __builtin_nontemporal_load(reinterpret_cast<const Raw *>(p)); }
    static __device__ __forceinline__ float elem(const Raw &r, int k) { return (float)r[k]; }
    // native half compare (v_cmp_gt_f16): half -> float is exact and 0.5 is a half, so this IS (float)x > 0.5f, minus
    // one v_cvt per pixel in the threshold / ballot step of the count kernels
    static __device__ __forceinline__ bool gt_half(const Raw &r, int k) { return r[k] > (_Float16)0.5f; }
    static __device__ __forceinline__ void loadv(const f16_t *p, float (&v)[8]) {
        half8u t = __builtin_nontemporal_load(reinterpret_cast<const half8u *>(p));
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (float)t[k];
    }
    template <bool NT = true> static __device__ __forceinline__ void load4(const f16_t *p, float (&v)[4]) {
        const half4u *q = reinterpret_cast<const half4u *>(p);
        half4u t;
        if (NT) t = __builtin_nontemporal_load(q);
        else t = *q;
        v[0] = (float)t.x; v[1] = (float)t.y; v[2] = (float)t.z; v[3] = (float)t.w;
    }
    static __device__ __forceinline__ float load1(const f16_t *p) { return (float)p->v; }
};
template <> struct MaskIO<bf16_t> {
    static constexpr int kVec = 8;
    typedef uint4u Raw;
    static __device__ __forceinline__ Raw load_raw(const bf16_t *p) { return __builtin_nontemporal_load(reinterpret_cast<const Raw *>(p)); }
    static __device__ __forceinline__ float elem(const Raw &r, int k) {
        return __uint_as_float((k & 1) ? (r[k >> 1] & 0xFFFF0000u) : (r[k >> 1] << 16));
    }
    static __device__ __forceinline__ bool gt_half(const Raw &r, int k) { return elem(r, k) > 0.5f; }
    static __device__ __forceinline__ void loadv(const bf16_t *p, float (&v)[8]) {
        uint4u t = __builtin_nontemporal_load(reinterpret_cast<const uint4u *>(p));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[2 * k] = __uint_as_float(t[k] << 16);
            v[2 * k + 1] = __uint_as_float(t[k] & 0xFFFF0000u);
        }
    }
    template <bool NT = true> static __device__ __forceinline__ void load4(const bf16_t *p, float (&v)[4]) {
        const uint2u *q = reinterpret_cast<const uint2u *>(p);
        uint2u t;
        if (NT) t = __builtin_nontemporal_load(q);
        else t = *q;
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xFFFF0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xFFFF0000u);
    }
    static __device__ __forceinline__ float load1(const bf16_t *p) { return __uint_as_float(((uint32_t)p->v) << 16); }
};

}  // namespace dmm

// ---- zero fill as a KERNEL ------------------------------------------------------------------
// The library's tables (count tables, dRb) are cleared by this launch, not by hipMemsetAsync.  A memset captured into a HIP
// graph becomes a memset NODE, and on this stack (ROCm 7.0.2, torch 2.10 hipGraph capture) a replayed graph did not order
// the kernel node that follows it behind that node: the count kernel of a replayed frame step accumulated onto the previous
// step's tables (found by tests/test_gpu_video.py::test_frame_loop_equals_an_oracle_computed_clip -- only on the path that
// clears with a memset, i.e. feature widths the fused similarity kernel does not take; eager launches were never affected).
// A kernel node is ordered like every other kernel of the chain.  bytes must be a multiple of 4 (all tables are 32-bit).
namespace dmm {
static __global__ __launch_bounds__(256) void zero_words_kernel(uint32_t *__restrict__ p, size_t words) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += stride) p[i] = 0u;
}
}  // namespace dmm

// ---- host side ------------------------------------------------------------------------------
namespace dmm {
void set_last_hip_error(int e);
void note_launch();                  // dmm_launch_count (diagnostic): one relaxed atomic increment per enqueued kernel
inline hipError_t zero_async(void *p, size_t bytes, hipStream_t stream) {
    const size_t words = bytes / 4;
    if (words == 0) return hipSuccess;
    size_t blocks = (words + 256 * 4 - 1) / (256 * 4);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (uint32_t *)p, words);
    note_launch();
    return hipGetLastError();
}
inline int check_launch() {
    note_launch();
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_hip_error((int)e);
        return DMM_ERR_LAUNCH;
    }
    return DMM_OK;
}
#define DMM_HIP_TRY(expr)                                  \
    do {                                                   \
        hipError_t _e = (expr);                            \
        if (_e != hipSuccess) {                            \
            ::dmm::set_last_hip_error((int)_e);            \
            return DMM_ERR_LAUNCH;                         \
        }                                                  \
    } while (0)

// ---- XCD-aware workgroup -> (frame, range) mapping -------------------------------------------
// Workgroups are dispatched round-robin over the 8 XCDs by linear id.  With the plain (range = blockIdx.x, frame =
// blockIdx.y) mapping the chunk ranges of one frame land on different XCDs, so the 128-byte lines neighbouring ranges
// share (plane rows are only 4-byte aligned) are fetched into several L2s.  Remapped, every complete group of 8 frames
// gives each XCD ONE whole frame: measured +2 % on the cost launches (6.65 -> 6.79 TB/s at 512 frames).
__device__ __forceinline__ void xcd_frame_range(int xcd_remap, int &b, int &range) {
    b = blockIdx.y;
    range = blockIdx.x;
    if (xcd_remap && (int)blockIdx.y < (int)(gridDim.y & ~7u)) {
        const int splits = gridDim.x;
        const int id = blockIdx.x + splits * blockIdx.y;
        const int grp = id / (8 * splits), within = id - grp * 8 * splits;
        b = grp * 8 + (within & 7);
        range = within >> 3;
    }
}
// the same for a workgroup that is number (bx, by) of a (gx, gy) sub-grid of its launch (the small-batch front kernel)
__device__ __forceinline__ void xcd_frame_range(int xcd_remap, int &b, int &range, int bx, int by, int gx, int gy) {
    b = by;
    range = bx;
    if (xcd_remap && by < (gy & ~7)) {
        const int splits = gx;
        const int id = bx + splits * by;
        const int grp = id / (8 * splits), within = id - grp * 8 * splits;
        b = grp * 8 + (within & 7);
        range = within >> 3;
    }
}


}  // namespace dmm
