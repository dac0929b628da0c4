// What a device-wide barrier costs on the MI355X: a persistent kernel of `wgs` workgroups (all resident) passes `n`
// barriers built from one atomic counter in HBM / L2 (arrive: atomicAdd + threadfence; wait: spin on a volatile load).
// The number bounds what a persistent multi-layer convolution kernel would pay between layers.
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ __launch_bounds__(256) void barrier_kernel(unsigned *counter, int n, float *sink) {
    float acc = 0.f;
    for (int i = 0; i < n; ++i) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            atomicAdd(counter, 1u);
            const unsigned target = (unsigned)(i + 1) * gridDim.x;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        acc += 1.f;
    }
    if (acc == 12345.f) sink[blockIdx.x] = acc;
}
extern "C" __attribute__((visibility("default"))) int probe(unsigned *counter, int wgs, int n, float *sink, void *stream) {
    hipMemsetAsync(counter, 0, 4, (hipStream_t)stream);
    hipLaunchKernelGGL(barrier_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, counter, n, sink);
    return (int)hipGetLastError();
}
