// Access-pattern probe for the train-mode mix (union kernel): how fast can 48 planes of a frame be streamed (and 10 rows
// written) with (A) every wave taking 1 KiB of each plane, 8 planes in flight, vs (C) every wave taking a 4 KiB run of ONE
// plane at a time (what an LDS-staged version would issue).  No arithmetic beyond keeping the loads alive.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

template <int UL, bool WRITE>
__global__ __launch_bounds__(256) void pat_a(const float *__restrict__ src, float *__restrict__ dst, int planes, int rows,
                                             int HW, int steps_per_wg, float *sink, int OS) {
    const float *fb = src + (int64_t)blockIdx.y * planes * HW;
    float *ob = dst + (int64_t)blockIdx.y * rows * OS;
    const int nsteps = HW / 1024;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = blockIdx.x * steps_per_wg; s < min(nsteps, (int)(blockIdx.x + 1) * steps_per_wg); ++s) {
        const int x = (s * 256 + threadIdx.x) * 4;
        for (int p0 = 0; p0 < planes; p0 += UL) {
            f4u v[UL];
#pragma unroll
            for (int u = 0; u < UL; ++u)
                v[u] = __builtin_nontemporal_load(reinterpret_cast<const f4u *>(fb + (int64_t)min(p0 + u, planes - 1) * HW + x));
#pragma unroll
            for (int u = 0; u < UL; ++u) acc += v[u];
        }
        if (WRITE)
            for (int m = 0; m < rows; ++m) {
                f4u t = acc + (float)m;
                __builtin_nontemporal_store(t, reinterpret_cast<f4u *>(ob + (int64_t)m * OS + x));
            }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[blockIdx.x] = acc.x;
}

// wave w of the workgroup streams the 4 KiB run [s * 1024, s * 1024 + 1024) floats of plane p0 + w (4 x 16 B per lane),
// NB batches of 4 planes in flight
template <int NB, bool WRITE>
__global__ __launch_bounds__(256) void pat_c(const float *__restrict__ src, float *__restrict__ dst, int planes, int rows,
                                             int HW, int steps_per_wg, float *sink, int OS) {
    const float *fb = src + (int64_t)blockIdx.y * planes * HW;
    float *ob = dst + (int64_t)blockIdx.y * rows * OS;
    const int nsteps = HW / 1024;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = blockIdx.x * steps_per_wg; s < min(nsteps, (int)(blockIdx.x + 1) * steps_per_wg); ++s) {
        for (int p0 = 0; p0 < planes; p0 += 4 * NB) {
            f4u v[NB][4];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int p = min(p0 + 4 * nb + wave, planes - 1);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    v[nb][j] = __builtin_nontemporal_load(reinterpret_cast<const f4u *>(fb + (int64_t)p * HW + s * 1024 + j * 256 + lane * 4));
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc += v[nb][j];
        }
        if (WRITE) {
            const int x = (s * 256 + threadIdx.x) * 4;
            for (int m = 0; m < rows; ++m) {
                f4u t = acc + (float)m;
                __builtin_nontemporal_store(t, reinterpret_cast<f4u *>(ob + (int64_t)m * OS + x));
            }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[blockIdx.x] = acc.x;
}

extern "C" __attribute__((visibility("default"))) int probe(int which, int write, const float *src, float *dst, int B,
                                                            int planes, int rows, int HW, int steps_per_wg, float *sink,
                                                            void *stream, int OS) {
    const int nsteps = HW / 1024;
    dim3 grid((nsteps + steps_per_wg - 1) / steps_per_wg, B);
    hipStream_t s = (hipStream_t)stream;
#define L(K) hipLaunchKernelGGL(K, grid, dim3(256), 0, s, src, dst, planes, rows, HW, steps_per_wg, sink, OS)
    if (which == 0) { if (write) L((pat_a<8, true>)); else L((pat_a<8, false>)); }
    else if (which == 1) { if (write) L((pat_a<4, true>)); else L((pat_a<4, false>)); }
    else if (which == 2) { if (write) L((pat_c<1, true>)); else L((pat_c<1, false>)); }
    else if (which == 3) { if (write) L((pat_c<2, true>)); else L((pat_c<2, false>)); }
    else { if (write) L((pat_a<16, true>)); else L((pat_a<16, false>)); }
    return (int)hipGetLastError();
}
