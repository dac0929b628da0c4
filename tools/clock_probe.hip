// What does ONE resident wave get from the chip?  Measures, inside a kernel, the shader clock actually applied to a
// 1-wave launch (s_memtime ticks per s_memrealtime tick, the latter a fixed 100 MHz) and the cost in shader cycles of
//   (a) a dependent v_add_f32 chain, (b) independent v_add_f32 ops, (c) a dependent DPP row_shl add chain,
//   (d) an LDS write -> read round trip, (e) v_readlane -> VALU use.
// Used to price the one-wave solver's sweep against its dependent chain (LABLOG round 4).
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void clock_probe_kernel(long long *out, float *sink, int n) {
    __shared__ float lds[256];
    const int lane = threadIdx.x;
    float a = (float)lane, b = 1.0f + sink[0];
    long long r0 = wall_clock64(), c0 = clock64();
    // (a) dependent adds
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) a = a + b;
    }
    long long t1 = clock64();
    // (b) 8 independent chains
    float x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
    long long t2 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 2; ++k) { x0 += b; x1 += b; x2 += b; x3 += b; x4 += b; x5 += b; x6 += b; x7 += b; }
    }
    long long t3 = clock64();
    a = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    // (c) dependent DPP adds
    long long t4 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
            a = a + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x101, 0xF, 0xF, false));
    }
    long long t5 = clock64();
    // (d) LDS write -> read round trips (dependent)
    long long t6 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lds[lane] = a;
            __builtin_amdgcn_wave_barrier();
            a = lds[(lane + 1) & 63] + b;
        }
    }
    long long t7 = clock64();
    // (e) readlane -> VALU (dependent)
    long long t8 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            a = a + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), (k * 8) & 63));
    }
    long long t9 = clock64();
    long long r1 = wall_clock64(), c1 = clock64();
    sink[1 + lane] = a;
    if (lane == 0 && blockIdx.x == 0) {
        out[0] = r1 - r0; out[1] = c1 - c0;
        out[2] = t1 - t0; out[3] = t3 - t2; out[4] = t5 - t4; out[5] = t7 - t6; out[6] = t9 - t8;
    }
}

// spin kernel to keep the rest of the chip busy (power state probe)
__global__ void busy_kernel(float *sink, int n) {
    float a = threadIdx.x, b = sink[0] + 1.0f;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k) a = __builtin_fmaf(a, b, 1.0f);
    }
    sink[128 + (blockIdx.x * blockDim.x + threadIdx.x) % 512] = a;
}

extern "C" int probe(long long *out, float *sink, int n, void *stream) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out, sink, n);
    return (int)hipGetLastError();
}
extern "C" int busy(float *sink, int n, int wgs, void *stream) {
    hipLaunchKernelGGL(busy_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, sink, n);
    return (int)hipGetLastError();
}
