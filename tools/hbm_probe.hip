// hbm_probe.hip -- read-only and copy streaming probes to calibrate the HBM ceiling the cost / mix kernels
// are priced against (diagnostic tool; not part of libdmm_match.so).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/hbm_probe.hip -o tools/libhbm_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f4 __attribute__((ext_vector_type(4)));

// every workgroup streams `per_wg` contiguous bytes (multiple of 256 * 16 * U)
template <int U>
__global__ __launch_bounds__(256) void read_kernel(const f4 *__restrict__ src, int64_t vec_per_wg, float *sink) {
    const f4 *p = src + (int64_t)blockIdx.x * vec_per_wg + threadIdx.x;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int64_t i = 0; i < vec_per_wg; i += 256 * U) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + i + u * 256);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[blockIdx.x] = acc.x;
}

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
// plain (cached) loads through a 4-byte-aligned vector type: what MaskIO<float>::load_raw issues
template <int U>
__global__ __launch_bounds__(256) void read_plain_kernel(const float *__restrict__ src, int64_t vec_per_wg, float *sink) {
    const f4u *p = reinterpret_cast<const f4u *>(src) + (int64_t)blockIdx.x * vec_per_wg + threadIdx.x;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int64_t i = 0; i < vec_per_wg; i += 256 * U) {
        f4u v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[blockIdx.x] = acc.x;
}

template <int U>
__global__ __launch_bounds__(256) void copy_kernel(const f4 *__restrict__ src, f4 *__restrict__ dst, int64_t vec_per_wg) {
    const int64_t base = (int64_t)blockIdx.x * vec_per_wg + threadIdx.x;
    for (int64_t i = 0; i < vec_per_wg; i += 256 * U) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(src + base + i + u * 256);
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u], dst + base + i + u * 256);
    }
}

extern "C" __attribute__((visibility("default"))) int probe_read(const void *src, int64_t bytes, int wgs, int unroll,
                                                                 float *sink, void *stream) {
    const int64_t vec_per_wg = bytes / 16 / wgs;
    if (unroll == 4) hipLaunchKernelGGL(read_kernel<4>, dim3(wgs), dim3(256), 0, (hipStream_t)stream, (const f4 *)src, vec_per_wg, sink);
    else if (unroll == 8) hipLaunchKernelGGL(read_kernel<8>, dim3(wgs), dim3(256), 0, (hipStream_t)stream, (const f4 *)src, vec_per_wg, sink);
    else hipLaunchKernelGGL(read_kernel<1>, dim3(wgs), dim3(256), 0, (hipStream_t)stream, (const f4 *)src, vec_per_wg, sink);
    return (int)hipGetLastError();
}

// the cost kernel's access pattern without its arithmetic: frame b = `planes` planes of HW floats (HW odd -> 4-byte
// aligned plane bases); a wave takes 1024-pixel chunks, visiting every plane of the chunk with U planes in flight
template <int U>
__global__ __launch_bounds__(256) void read_planes_kernel(const float *__restrict__ src, int planes, int HW,
                                                          int chunks_per_wg, float *sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float *fb = src + (int64_t)blockIdx.y * planes * HW;
    const int full = HW / 1024;
    const int c_begin = blockIdx.x * chunks_per_wg, c_end = min(full, c_begin + chunks_per_wg);
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int c = c_begin + wave; c < c_end; c += 4) {
        const float *x = fb + c * 1024 + lane * 4;
        for (int p0 = 0; p0 < planes; p0 += U) {
            f4u v[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int p = p0 + u < planes ? p0 + u : planes - 1;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    v[u][j] = __builtin_nontemporal_load(reinterpret_cast<const f4u *>(x + (int64_t)p * HW + j * 256));
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc += v[u][j];
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[blockIdx.x] = acc.x;
}

// same, but a wave visit takes RUN consecutive 4 KiB blocks of the plane (run length RUN * 4 KiB)
template <int RUN>
__global__ __launch_bounds__(256) void read_planes_run_kernel(const float *__restrict__ src, int planes, int HW,
                                                              int chunks_per_wg, float *sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float *fb = src + (int64_t)blockIdx.y * planes * HW;
    const int full = HW / (1024 * RUN);
    const int c_begin = blockIdx.x * chunks_per_wg, c_end = min(full, c_begin + chunks_per_wg);
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int c = c_begin + wave; c < c_end; c += 4) {
        const float *x = fb + (int64_t)c * 1024 * RUN + lane * 4;
        for (int p = 0; p < planes; ++p) {
            f4u v[RUN][4];
#pragma unroll
            for (int r = 0; r < RUN; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    v[r][j] = __builtin_nontemporal_load(reinterpret_cast<const f4u *>(x + (int64_t)p * HW + r * 1024 + j * 256));
#pragma unroll
            for (int r = 0; r < RUN; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc += v[r][j];
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[blockIdx.x] = acc.x;
}

// config-5 geometry: a wave visit takes ONE 2 KiB run (two 16-byte loads per lane) of the plane -- what a 1024-pixel
// chunk of a 16-bit plane is -- with UNR planes in flight
template <int UNR>
__global__ __launch_bounds__(256) void read_planes_half_kernel(const float *__restrict__ src, int planes, int HW,
                                                               int chunks_per_wg, float *sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float *fb = src + (int64_t)blockIdx.y * planes * HW;
    const int full = HW / 512;
    const int c_begin = blockIdx.x * chunks_per_wg, c_end = min(full, c_begin + chunks_per_wg);
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int c = c_begin + wave; c < c_end; c += 4) {
        const float *x = fb + (int64_t)c * 512 + lane * 4;
        for (int p0 = 0; p0 < planes; p0 += UNR) {
            f4u v[UNR][2];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int p = p0 + u < planes ? p0 + u : planes - 1;
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    v[u][j] = __builtin_nontemporal_load(reinterpret_cast<const f4u *>(x + (int64_t)p * HW + j * 256));
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc += v[u][j];
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[blockIdx.x] = acc.x;
}

// the cost-kernel pattern with every 4 KiB run moved down to its 128-byte line boundary (what per-plane aligned loads
// + funnel-shifted ballot words would read): is the odd plane size or the plane pattern the limit?
template <int ALIGN>
__global__ __launch_bounds__(256) void read_planes_aligned_kernel(const float *__restrict__ src, int planes, int HW,
                                                                  int chunks_per_wg, float *sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float *fb = src + (int64_t)blockIdx.y * planes * HW;
    const int full = HW / 1024;
    const int c_begin = blockIdx.x * chunks_per_wg, c_end = min(full, c_begin + chunks_per_wg);
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int c = c_begin + wave; c < c_end; c += 4) {
        for (int p0 = 0; p0 < planes; p0 += 2) {
            f4 v[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int p = p0 + u < planes ? p0 + u : planes - 1;
                const uintptr_t a = reinterpret_cast<uintptr_t>(fb + (int64_t)p * HW + c * 1024) & ~(uintptr_t)(ALIGN - 1);
                const f4 *x = reinterpret_cast<const f4 *>(a) + lane;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[u][j] = __builtin_nontemporal_load(x + j * 64);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc += v[u][j];
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[blockIdx.x] = acc.x;
}

extern "C" __attribute__((visibility("default"))) int probe_read_planes_aligned(const void *src, int B, int planes, int HW,
                                                                                int wgs, int align, float *sink, void *stream) {
    const int nchunks = HW / 1024;
    int splits = (wgs + B - 1) / B;
    if (splits > (nchunks + 3) / 4) splits = (nchunks + 3) / 4;
    const int cpw = (nchunks + splits - 1) / splits;
    splits = (nchunks + cpw - 1) / cpw;
    dim3 grid(splits, B);
    if (align == 16) hipLaunchKernelGGL(read_planes_aligned_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, (const float *)src, planes, HW, cpw, sink);
    else if (align == 64) hipLaunchKernelGGL(read_planes_aligned_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, (const float *)src, planes, HW, cpw, sink);
    else hipLaunchKernelGGL(read_planes_aligned_kernel<128>, grid, dim3(256), 0, (hipStream_t)stream, (const float *)src, planes, HW, cpw, sink);
    return (int)hipGetLastError();
}

// compact footprint: one workgroup = ONE 1024-pixel chunk of one frame, its 4 waves split the planes
// (grid = (chunks, B)): the resident workgroups cover ~16 frames instead of ~64
__global__ __launch_bounds__(256) void read_planes_split_kernel(const float *__restrict__ src, int planes, int HW,
                                                                float *sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float *x = src + (int64_t)blockIdx.y * planes * HW + blockIdx.x * 1024 + lane * 4;
    const int per = (planes + 3) / 4;
    const int p_begin = wave * per, p_end = min(planes, p_begin + per);
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int p0 = p_begin; p0 < p_end; p0 += 2) {
        f4u v[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int p = p0 + u < p_end ? p0 + u : p_end - 1;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                v[u][j] = __builtin_nontemporal_load(reinterpret_cast<const f4u *>(x + (int64_t)p * HW + j * 256));
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += v[u][j];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[blockIdx.x] = acc.x;
}

extern "C" __attribute__((visibility("default"))) int probe_read_planes_split(const void *src, int B, int planes, int HW,
                                                                              float *sink, void *stream) {
    hipLaunchKernelGGL(read_planes_split_kernel, dim3(HW / 1024, B), dim3(256), 0, (hipStream_t)stream, (const float *)src,
                       planes, HW, sink);
    return (int)hipGetLastError();
}

extern "C" __attribute__((visibility("default"))) int probe_read_planes_run(const void *src, int B, int planes, int HW,
                                                                            int wgs, int run, float *sink, void *stream) {
    const int nchunks = HW / (1024 * run);
    int splits = (wgs + B - 1) / B;
    if (splits > (nchunks + 3) / 4) splits = (nchunks + 3) / 4;
    const int cpw = (nchunks + splits - 1) / splits;
    splits = (nchunks + cpw - 1) / cpw;
    dim3 grid(splits, B);
    if (run == 1) hipLaunchKernelGGL(read_planes_run_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, (const float *)src, planes, HW, cpw, sink);
    else if (run == 2) hipLaunchKernelGGL(read_planes_run_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, (const float *)src, planes, HW, cpw, sink);
    else hipLaunchKernelGGL(read_planes_run_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, (const float *)src, planes, HW, cpw, sink);
    return (int)hipGetLastError();
}

extern "C" __attribute__((visibility("default"))) int probe_read_planes(const void *src, int B, int planes, int HW,
                                                                        int wgs, int unroll, float *sink, void *stream) {
    const int nchunks = HW / 1024;
    int splits = (wgs + B - 1) / B;
    if (splits > (nchunks + 3) / 4) splits = (nchunks + 3) / 4;
    const int cpw = (nchunks + splits - 1) / splits;
    splits = (nchunks + cpw - 1) / cpw;
    dim3 grid(splits, B);
    if (unroll == 1) hipLaunchKernelGGL(read_planes_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, (const float *)src, planes, HW, cpw, sink);
    else if (unroll == 2) hipLaunchKernelGGL(read_planes_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, (const float *)src, planes, HW, cpw, sink);
    else hipLaunchKernelGGL(read_planes_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, (const float *)src, planes, HW, cpw, sink);
    return (int)hipGetLastError();
}

extern "C" __attribute__((visibility("default"))) int probe_read_plain(const void *src, int64_t bytes, int wgs, int unroll,
                                                                       float *sink, void *stream) {
    const int64_t vec_per_wg = bytes / 16 / wgs;
    if (unroll == 4) hipLaunchKernelGGL(read_plain_kernel<4>, dim3(wgs), dim3(256), 0, (hipStream_t)stream, (const float *)src, vec_per_wg, sink);
    else hipLaunchKernelGGL(read_plain_kernel<1>, dim3(wgs), dim3(256), 0, (hipStream_t)stream, (const float *)src, vec_per_wg, sink);
    return (int)hipGetLastError();
}

extern "C" __attribute__((visibility("default"))) int probe_copy(const void *src, void *dst, int64_t bytes, int wgs,
                                                                 int unroll, void *stream) {
    const int64_t vec_per_wg = bytes / 16 / wgs;
    if (unroll == 4) hipLaunchKernelGGL(copy_kernel<4>, dim3(wgs), dim3(256), 0, (hipStream_t)stream, (const f4 *)src, (f4 *)dst, vec_per_wg);
    else hipLaunchKernelGGL(copy_kernel<1>, dim3(wgs), dim3(256), 0, (hipStream_t)stream, (const f4 *)src, (f4 *)dst, vec_per_wg);
    return (int)hipGetLastError();
}

extern "C" __attribute__((visibility("default"))) int probe_read_planes_half(const void *src, int B, int planes, int HW,
                                                                             int wgs, int unr, float *sink, void *stream) {
    const int nchunks = HW / 512;
    int splits = (wgs + B - 1) / B;
    if (splits > (nchunks + 3) / 4) splits = (nchunks + 3) / 4;
    const int cpw = (nchunks + splits - 1) / splits;
    splits = (nchunks + cpw - 1) / cpw;
    dim3 grid(splits, B);
    if (unr == 4) hipLaunchKernelGGL(read_planes_half_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, (const float *)src, planes, HW, cpw, sink);
    else hipLaunchKernelGGL(read_planes_half_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, (const float *)src, planes, HW, cpw, sink);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Skeleton of iou_counts_tl_kernel (config 5) with its ingredients switchable, to find what separates it from the
// bare read loop: MODE bit 0 = the thresholding / ballot / and / bcnt work against a parked 32-register tile,
// bit 1 = the LDS atomic per plane; HWh = plane size in HALVES (65025: odd, planes 2 bytes off; 65024: line aligned);
// dynamic LDS caps the workgroups per CU (40 KB -> 4 workgroups = 4 waves per SIMD, like the kernel's 128 VGPRs).
// Ping-pong groups of 4 planes x 2 KiB, 1024-pixel chunks, 4 waves per workgroup: the kernel's loop structure.
typedef _Float16 h8u __attribute__((ext_vector_type(8), aligned(2)));
template <int MODE>
__global__ __launch_bounds__(256) void tl_skeleton_kernel(const _Float16 *__restrict__ src, int planes, int HWh,
                                                           int chunks_per_wg, unsigned *sink) {
    extern __shared__ unsigned red[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const _Float16 *fb = src + (int64_t)blockIdx.y * planes * HWh;
    const int full = HWh / 1024;
    const int c_begin = blockIdx.x * chunks_per_wg, c_end = min(full, c_begin + chunks_per_wg);
    for (int i = threadIdx.x; i < 256 * 22; i += 256) red[i] = 0;
    __syncthreads();
    unsigned tot = 0;
    int tlo[16], thi[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { tlo[k] = 0x5a5a5a5a ^ (lane * 0x01010101) ^ k; thi[k] = 0xa5a5a5a5 ^ (lane * 0x10101010) ^ k; }
    for (int c = c_begin + wave; c < c_end; c += 4) {
        const _Float16 *x = fb + (int64_t)c * 1024 + lane * 8;
        auto load_group = [&](h8u (&v)[4][2], int p0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = p0 + u < planes ? p0 + u : planes - 1;
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    v[u][j] = __builtin_nontemporal_load(reinterpret_cast<const h8u *>(x + (int64_t)p * HWh + j * 512));
            }
        };
        auto count_group = [&](const h8u (&v)[4][2], int p0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (p0 + u < planes) {
                    unsigned a0 = 0, a1 = 0;
                    if (MODE & 1) {
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                const unsigned long long b = __ballot(v[u][j][k] > (_Float16)0.5f);
                                const int lo = tlo[8 * j + k] & (int)(unsigned)b, hi = thi[8 * j + k] & (int)(unsigned)(b >> 32);
                                asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a0) : "v"(lo));
                                asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a1) : "v"(hi));
                            }
                    } else {
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const unsigned *w = reinterpret_cast<const unsigned *>(&v[u][j]);
                            a0 |= w[0] | w[1];
                            a1 |= w[2] | w[3];
                        }
                    }
                    if (MODE & 2) { if (lane <= 20) atomicAdd(&red[(p0 + u) * 22 + lane], a0 + a1); }
                    else tot += a0 + a1;
                }
            }
        };
        h8u va[4][2], vb[4][2];
        load_group(va, 0);
        for (int p0 = 0; p0 < planes; p0 += 8) {
            if (p0 + 4 < planes) load_group(vb, p0 + 4);
            count_group(va, p0);
            if (p0 + 4 >= planes) break;
            if (p0 + 8 < planes) load_group(va, p0 + 8);
            count_group(vb, p0 + 4);
        }
    }
    __syncthreads();
    if (MODE & 2) tot += red[threadIdx.x];
    if (tot == 0x12345678u) sink[blockIdx.x] = tot;
}
extern "C" __attribute__((visibility("default"))) int probe_tl_skeleton(const void *src, int B, int planes, int HWh, int wgs,
                                                                        int mode, int lds_bytes, void *sink, void *stream) {
    const int nchunks = HWh / 1024;
    int splits = (wgs + B - 1) / B;
    if (splits > (nchunks + 3) / 4) splits = (nchunks + 3) / 4;
    const int cpw = (nchunks + splits - 1) / splits;
    splits = (nchunks + cpw - 1) / cpw;
    dim3 grid(splits, B);
    if (lds_bytes < 256 * 22 * 4) lds_bytes = 256 * 22 * 4;
#define TLS(M)                                                                                                      \
    case M:                                                                                                         \
        if (lds_bytes > 65536) hipFuncSetAttribute((const void *)tl_skeleton_kernel<M>,                             \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);          \
        hipLaunchKernelGGL(tl_skeleton_kernel<M>, grid, dim3(256), lds_bytes, (hipStream_t)stream,                  \
                           (const _Float16 *)src, planes, HWh, cpw, (unsigned *)sink);                              \
        break;
    switch (mode) { TLS(0) TLS(1) TLS(2) TLS(3) }
#undef TLS
    return (int)hipGetLastError();
}
