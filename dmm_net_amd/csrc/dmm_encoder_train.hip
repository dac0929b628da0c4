// dmm_encoder_train.hip -- training-mode BatchNorm (+ residual) (+ ReLU) of the encoder on gfx950, channels-last bf16.
//
// Reference: the conv -> BatchNorm -> ReLU stacks the trainer differentiates through: dmm/modules/base.py:43-54 (prop heads),
// model_encoder.py:137-146 (skip projections + bn), the torchvision Bottleneck / BasicBlock bodies reached through
// dmm/modules/vision.py:6-38 (out = relu(bn3(conv3(x)) + identity)), under train.py:296-307 (forward, loss.backward()).
// The stock path issues per layer and direction three MIOpen BatchNorm kernels + an add + a clamp (684 BatchNorm and 427
// elementwise launches of the ~2 100 of one ResNet-101 step at 12 x 255 x 448; 6.75 of 17.6 ms of device time,
// profiles/r06_cfg4_kernel_stats_autocast_nhwc.csv).  Here it is TWO launches each way:
//
//   forward    bn_stats:       per-channel sum(x), sum(x^2)                                 (reads x)
//              bn_apply:       y = act(x * scale + shift (+ residual)); running statistics  (reads x (+ residual), writes y)
//   backward   bn_bwd_reduce:  sum(g), sum(g * xhat),  g = dy * [y > 0]                      (reads dy, x (, y))
//              bn_bwd_dx:      dx = w * invstd * (g - mean(g) - xhat * mean(g * xhat));      (reads dy, x (, y), writes dx
//                              g itself = the residual branch's gradient; dweight, dbias       (+ dres))
//
// Layout: x [rows, C] bf16 (a channels-last activation viewed 2-D), C % 8 == 0 and C / 8 a divisor of 256 (every width of
// the ResNets and heads: 32 .. 2048).  One thread = 8 consecutive channels (one 16-byte load) of every (256 / (C/8))-th row
// of its workgroup's row range; fp32 arithmetic, one rounding.  Per-channel sums: per-thread fp32 accumulators -> one LDS
// fold per workgroup -> one global fp32 atomic per (workgroup, channel, statistic) into a [2, C] buffer the caller zeroed.
// (Atomic arrival order is free: the statistics of two runs may differ in their last bits, as MIOpen's do.)
// var = E[x^2] - mean^2 in fp32, clamped at 0: inputs carry 8 significant bits and post-convolution activations have
// |mean| / std of order 1, so the cancellation costs nothing a bf16 activation could show.
//
// Roofline: HBM / L2 streaming (3-4 passes forward, 5-7 backward over tensors of 1.4 .. 44 MB); the small ones are launch
// bound, which is why the whole step replays from HIP graphs (dmm_net_amd/train_encoder.py).
#include "dmm_common.h"

namespace dmm {

typedef uint32_t u32x4t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t bf16_round(float v) {
    const uint32_t u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

__device__ __forceinline__ void unpack8(const u32x4t v, float *f) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f[2 * k] = __uint_as_float(v[k] << 16);
        f[2 * k + 1] = __uint_as_float(v[k] & 0xFFFF0000u);
    }
}

__device__ __forceinline__ u32x4t pack8(const float *f) {
    u32x4t o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = bf16_round(f[2 * k]) | (bf16_round(f[2 * k + 1]) << 16);
    return o;
}

__device__ __forceinline__ void load8f(const float *p, float *f) {
    const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

// fold the 16 per-thread partials (two statistics x 8 channels) of the threads that share a channel group, then one
// atomic per (channel, statistic) of the workgroup.  red: [256][16] floats.  c8t = channel groups of this workgroup's tile
// (threads per row), cg0 = first channel group of the tile.
__device__ __forceinline__ void fold_and_add(const float *acc16, float *red, int c8t, int cg0, int C,
                                             float *__restrict__ out2C) {
    const int t = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 16; ++k) red[t * 16 + k] = acc16[k];
    __syncthreads();
    const int rpp = 256 / c8t;                           // threads (rows per pass) per channel group
    for (int j = t; j < 16 * c8t; j += 256) {            // j = cg * 16 + k
        float s = 0.0f;
        for (int r = 0; r < rpp; ++r) s += red[(r * c8t) * 16 + j];
        const int cg = j >> 4, k = j & 15;
        unsafeAtomicAdd(&out2C[(k >> 3) * C + (cg0 + cg) * 8 + (k & 7)], s);
    }
}

// rows of workgroup g: [g * per, min(rows, (g + 1) * per)), per a multiple of the rows one pass covers
__device__ __forceinline__ void row_range(int64_t rows, int rpp, int64_t &r0, int64_t &r1) {
    int64_t per = (rows + gridDim.x - 1) / gridDim.x;
    per = (per + rpp - 1) / rpp * rpp;
    r0 = (int64_t)blockIdx.x * per;
    r1 = r0 + per < rows ? r0 + per : rows;
}

// Statistics kernels: grid = (row groups, channel tiles of <= 256 channels).  Device-scope fp32 atomics run at ~50 G/s on
// this part (a launch of 2048 workgroups x 2C atomics took 24 us where the data pass takes 5), so the ROW groups are few --
// row groups x 2C <= 128 k atomics -- and wide layers get their parallelism from the channel tiles instead.
__global__ __launch_bounds__(256) void bn_stats_bf16_kernel(const uint16_t *__restrict__ x, int64_t rows, int c8,
                                                            float *__restrict__ stats) {
    __shared__ float red[256 * 16];
    const int c8t = c8 < 32 ? c8 : 32, cg0 = blockIdx.y * c8t;
    const int cg = cg0 + threadIdx.x % c8t, ro = threadIdx.x / c8t, rpp = 256 / c8t;
    int64_t r0, r1;
    row_range(rows, rpp, r0, r1);
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.0f;
    // (blockIdx.z = the statistics group: `rows` consecutive rows with their own [2, C] block)
    const u32x4t *xp = reinterpret_cast<const u32x4t *>(x) + (int64_t)blockIdx.z * rows * c8;
    stats += (int64_t)blockIdx.z * 2 * c8 * 8;
    // eight loads in flight, the tail included: a pass beyond the range reads the thread's first row again and counts as zeros
    // (a serial tail of up to seven dependent loads was half the kernel's time on the short tensors of layer3 / layer4)
    const u32x4t z = {0u, 0u, 0u, 0u};
    for (int64_t r = r0 + ro; r < r1; r += 8 * rpp) {
        u32x4t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t rr = r + u * rpp;
            v[u] = xp[(rr < r1 ? rr : r) * c8 + cg];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float f[8];
            unpack8(r + u * rpp < r1 ? v[u] : z, f);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc[k] += f[k];
                acc[8 + k] = __builtin_fmaf(f[k], f[k], acc[8 + k]);
            }
        }
    }
    fold_and_add(acc, red, c8t, cg0, c8 * 8, stats);
}

template <bool RES, bool RELU>
__global__ __launch_bounds__(256) void bn_apply_bf16_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ res,
                                                            int64_t rows, int c8, const float *__restrict__ stats,
                                                            const float *__restrict__ weight, const float *__restrict__ bias,
                                                            float *__restrict__ running_mean, float *__restrict__ running_var,
                                                            float momentum, float eps, uint16_t *__restrict__ y,
                                                            float *__restrict__ saved) {
    const int C = c8 * 8, cg = threadIdx.x % c8, ro = threadIdx.x / c8, rpp = 256 / c8;
    const int grp = blockIdx.z, groups = gridDim.z;       // statistics groups: `rows` consecutive rows each
    float s[8], q[8], w[8], b[8], scale[8], shift[8];
    load8f(stats + (int64_t)grp * 2 * C + cg * 8, s);
    load8f(stats + (int64_t)grp * 2 * C + C + cg * 8, q);
    load8f(weight + cg * 8, w);
    load8f(bias + cg * 8, b);
    const float inv_n = 1.0f / (float)rows;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float mean = s[k] * inv_n;
        float var = __builtin_fmaf(-mean, mean, q[k] * inv_n);
        var = var > 0.0f ? var : 0.0f;
        const float invstd = 1.0f / __builtin_sqrtf(var + eps);
        scale[k] = w[k] * invstd;
        shift[k] = __builtin_fmaf(-mean, scale[k], b[k]);
        if (blockIdx.x == 0 && ro == 0) {
            const int c = cg * 8 + k;
            saved[(int64_t)grp * 2 * C + c] = mean;
            saved[(int64_t)grp * 2 * C + C + c] = invstd;
        }
    }
    if (running_mean && blockIdx.x == 0 && grp == 0 && ro == 0) {
        // torch: running = (1 - m) * running + m * batch with the unbiased variance -- once per group, in group order (what
        // `groups` calls of the module, one per group, leave behind)
        for (int gq = 0; gq < groups; ++gq) {
            load8f(stats + (int64_t)gq * 2 * C + cg * 8, s);
            load8f(stats + (int64_t)gq * 2 * C + C + cg * 8, q);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int c = cg * 8 + k;
                const float mean = s[k] * inv_n;
                float var = __builtin_fmaf(-mean, mean, q[k] * inv_n);
                var = var > 0.0f ? var : 0.0f;
                const float unb = rows > 1 ? var * ((float)rows / (float)(rows - 1)) : var;
                running_mean[c] = __builtin_fmaf(momentum, mean - running_mean[c], running_mean[c]);
                running_var[c] = __builtin_fmaf(momentum, unb - running_var[c], running_var[c]);
            }
        }
    }
    int64_t r0, r1;
    row_range(rows, rpp, r0, r1);
    const int64_t gofs = (int64_t)grp * rows * c8;
    const u32x4t *xp = reinterpret_cast<const u32x4t *>(x) + gofs, *rp = reinterpret_cast<const u32x4t *>(res) + gofs;
    u32x4t *yp = reinterpret_cast<u32x4t *>(y) + gofs;
    for (int64_t r = r0 + ro; r < r1; r += rpp) {
        const int64_t i = r * c8 + cg;
        float f[8], g[8];
        unpack8(xp[i], f);
        if (RES) unpack8(rp[i], g);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = __builtin_fmaf(f[k], scale[k], shift[k]);
            if (RES) v += g[k];
            if (RELU) v = v > 0.0f ? v : 0.0f;
            f[k] = v;
        }
        yp[i] = pack8(f);
    }
}

// RELU: 0 = none, 1 = the mask from the rounded output y, 2 = the mask recomputed from x (no residual in front of the ReLU:
// y > 0 exactly when x * scale + shift > 0, the forward's own fma with the forward's own scale / shift -- one plane less to read)
template <int RELU>
__global__ __launch_bounds__(256) void bn_bwd_reduce_bf16_kernel(const uint16_t *__restrict__ dy, const uint16_t *__restrict__ dy2,
                                                                 const uint16_t *__restrict__ x,
                                                                 const uint16_t *__restrict__ y, int64_t rows, int c8,
                                                                 const float *__restrict__ saved, const float *__restrict__ weight,
                                                                 const float *__restrict__ bias, float *__restrict__ sums) {
    __shared__ float red[256 * 16];
    const int C = c8 * 8, c8t = c8 < 32 ? c8 : 32, cg0 = blockIdx.y * c8t;
    const int cg = cg0 + threadIdx.x % c8t, ro = threadIdx.x / c8t, rpp = 256 / c8t;
    saved += (int64_t)blockIdx.z * 2 * C;                 // (blockIdx.z = the statistics group)
    sums += (int64_t)blockIdx.z * 2 * C;
    float mean[8], invstd[8], scale[8], shift[8];
    load8f(saved + cg * 8, mean);
    load8f(saved + C + cg * 8, invstd);
    if (RELU == 2) {
        load8f(weight + cg * 8, scale);
        load8f(bias + cg * 8, shift);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            scale[k] = scale[k] * invstd[k];
            shift[k] = __builtin_fmaf(-mean[k], scale[k], shift[k]);
        }
    }
    int64_t r0, r1;
    row_range(rows, rpp, r0, r1);
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.0f;
    const int64_t gofs = (int64_t)blockIdx.z * rows * c8;
    const u32x4t *dp = reinterpret_cast<const u32x4t *>(dy) + gofs, *xp = reinterpret_cast<const u32x4t *>(x) + gofs,
                 *yp = reinterpret_cast<const u32x4t *>(y) + gofs;
    // dy2 (may be null): the gradient is dy + dy2 -- the layer's output went to two consumers (the next block's first
    // convolution and its identity branch) and their gradients are added HERE, in fp32, instead of by a launch of their own
    const u32x4t *dp2 = dy2 ? reinterpret_cast<const u32x4t *>(dy2) + gofs : nullptr;
    // U rows (12-16 loads) in flight, the tail included: a pass beyond the range reads the thread's first row again with a zero
    // gradient (contributes nothing to either sum)
    constexpr int U = RELU == 1 ? 4 : 8;
    const u32x4t z = {0u, 0u, 0u, 0u};
    for (int64_t r = r0 + ro; r < r1; r += U * rpp) {
        u32x4t vd[U], vx[U], vy[U], ve[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t rr = r + u * rpp;
            const int64_t i = (rr < r1 ? rr : r) * c8 + cg;
            vd[u] = dp[i];
            ve[u] = dp2 ? dp2[i] : z;
            vx[u] = xp[i];
            if (RELU == 1) vy[u] = yp[i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float g[8], f[8], o[8], g2[8];
            unpack8(r + u * rpp < r1 ? vd[u] : z, g);
            unpack8(r + u * rpp < r1 ? ve[u] : z, g2);
#pragma unroll
            for (int k = 0; k < 8; ++k) g[k] += g2[k];
            unpack8(vx[u], f);
            if (RELU == 1) unpack8(vy[u], o);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float gk = (RELU == 0 || (RELU == 1 ? o[k] > 0.0f : __builtin_fmaf(f[k], scale[k], shift[k]) > 0.0f)) ? g[k] : 0.0f;
                acc[k] += gk;
                acc[8 + k] = __builtin_fmaf(gk, (f[k] - mean[k]) * invstd[k], acc[8 + k]);
            }
        }
    }
    fold_and_add(acc, red, c8t, cg0, C, sums);
}

template <int RELU, bool DRES>
__global__ __launch_bounds__(256) void bn_bwd_dx_bf16_kernel(const uint16_t *__restrict__ dy, const uint16_t *__restrict__ dy2,
                                                             const uint16_t *__restrict__ x,
                                                             const uint16_t *__restrict__ y, int64_t rows, int c8,
                                                             const float *__restrict__ saved, const float *__restrict__ weight,
                                                             const float *__restrict__ bias,
                                                             const float *__restrict__ sums, uint16_t *__restrict__ dx,
                                                             uint16_t *__restrict__ dres, float *__restrict__ dweight,
                                                             float *__restrict__ dbias) {
    const int C = c8 * 8, cg = threadIdx.x % c8, ro = threadIdx.x / c8, rpp = 256 / c8;
    const int grp = blockIdx.z, groups = gridDim.z;       // statistics groups: `rows` consecutive rows each
    float mean[8], invstd[8], w[8], sg[8], sgx[8], a[8], mg[8], mgx[8], shift[8];
    load8f(saved + (int64_t)grp * 2 * C + cg * 8, mean);
    load8f(saved + (int64_t)grp * 2 * C + C + cg * 8, invstd);
    load8f(weight + cg * 8, w);
    load8f(sums + (int64_t)grp * 2 * C + cg * 8, sg);
    load8f(sums + (int64_t)grp * 2 * C + C + cg * 8, sgx);
    if (RELU == 2) load8f(bias + cg * 8, shift);
    const float inv_n = 1.0f / (float)rows;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        a[k] = w[k] * invstd[k];
        if (RELU == 2) shift[k] = __builtin_fmaf(-mean[k], a[k], shift[k]);     // (a = the forward's scale)
        mg[k] = sg[k] * inv_n;
        mgx[k] = sgx[k] * inv_n;
    }
    if (blockIdx.x == 0 && grp == 0 && ro == 0) {         // the parameters' gradients: the groups' sums added in group order
        float tg[8], tgx[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) tg[k] = tgx[k] = 0.0f;
        for (int gq = 0; gq < groups; ++gq) {
            float u[8], v[8];
            load8f(sums + (int64_t)gq * 2 * C + cg * 8, u);
            load8f(sums + (int64_t)gq * 2 * C + C + cg * 8, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                tg[k] += u[k];
                tgx[k] += v[k];
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            dweight[cg * 8 + k] = tgx[k];
            dbias[cg * 8 + k] = tg[k];
        }
    }
    int64_t r0, r1;
    row_range(rows, rpp, r0, r1);
    const int64_t gofs = (int64_t)grp * rows * c8;
    const u32x4t *dp = reinterpret_cast<const u32x4t *>(dy) + gofs, *xp = reinterpret_cast<const u32x4t *>(x) + gofs,
                 *yp = reinterpret_cast<const u32x4t *>(y) + gofs;
    u32x4t *op = reinterpret_cast<u32x4t *>(dx) + gofs, *rp = reinterpret_cast<u32x4t *>(dres) + gofs;
    const u32x4t *dp2 = dy2 ? reinterpret_cast<const u32x4t *>(dy2) + gofs : nullptr;
    for (int64_t r = r0 + ro; r < r1; r += rpp) {
        const int64_t i = r * c8 + cg;
        float g[8], f[8], o[8];
        unpack8(dp[i], g);
        if (dp2) {
            float g2[8];
            unpack8(dp2[i], g2);
#pragma unroll
            for (int k = 0; k < 8; ++k) g[k] += g2[k];
        }
        unpack8(xp[i], f);
        if (RELU == 1) unpack8(yp[i], o);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float gk = (RELU == 0 || (RELU == 1 ? o[k] > 0.0f : __builtin_fmaf(f[k], a[k], shift[k]) > 0.0f)) ? g[k] : 0.0f;
            g[k] = gk;
            const float xhat = (f[k] - mean[k]) * invstd[k];
            f[k] = a[k] * ((gk - mg[k]) - xhat * mgx[k]);
        }
        op[i] = pack8(f);
        if (DRES) rp[i] = pack8(g);                      // (g = dy under the mask: exact in bf16)
    }
}

// 3x3 weights of a whole segment in ONE launch: for every entry of a device table, the fp32 master [co, ci, 3, 3] becomes the bf16
// channels-last weight dst[co][kh][kw][ci] and (dstT != null) the weight of the convolution that computes the DATA gradient of a
// stride 1 / padding 1 convolution, dstT[ci][2 - kh][2 - kw][co].  One workgroup per 32 x 32 (co, ci) tile: the 288 floats of a
// (co, 32 ci) run are contiguous in the master, the 32-channel runs of both outputs are 64-byte segments.
struct WPrep {
    const float *src;
    uint16_t *dst, *dstT;
    int co, ci;
    int64_t tile0;                        // first workgroup of this entry
};

__global__ __launch_bounds__(256) void wprep3x3_bf16_kernel(const WPrep *__restrict__ tab, int n) {
    constexpr int SO = 290;               // LDS row of one co: 288 values + 2 (odd dword stride: the transposed read is conflict free)
    __shared__ uint16_t s[32 * SO];
    const int64_t b = blockIdx.x;
    int k = 0;
    for (int i = 1; i < n; ++i) k = tab[i].tile0 <= b ? i : k;
    const WPrep e = tab[k];
    const int t = (int)(b - e.tile0), ct = e.ci / 32, o0 = (t / ct) * 32, c0 = (t % ct) * 32;
    for (int j = threadIdx.x; j < 32 * 288; j += 256) {
        const int o = j / 288, r = j - o * 288;
        s[o * SO + r] = (uint16_t)bf16_round(e.src[((int64_t)(o0 + o) * e.ci + c0) * 9 + r]);
    }
    __syncthreads();
    // 16-byte stores: 8 consecutive channels per thread (2-byte stores made this kernel 25 us per segment of a ResNet-101)
    for (int v = threadIdx.x; v < 32 * 36; v += 256) {
        const int c8 = (v & 3) * 8, tt = (v >> 2) % 9, o = v / 36;
        u32x4t w;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            w[k] = (uint32_t)s[o * SO + (c8 + 2 * k) * 9 + tt] | ((uint32_t)s[o * SO + (c8 + 2 * k + 1) * 9 + tt] << 16);
        *reinterpret_cast<u32x4t *>(e.dst + ((int64_t)(o0 + o) * 9 + tt) * e.ci + c0 + c8) = w;
    }
    if (!e.dstT) return;
    for (int v = threadIdx.x; v < 32 * 36; v += 256) {
        const int o8 = (v & 3) * 8, tt = (v >> 2) % 9, c = v / 36;
        u32x4t w;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            w[k] = (uint32_t)s[(o8 + 2 * k) * SO + c * 9 + tt] | ((uint32_t)s[(o8 + 2 * k + 1) * SO + c * 9 + tt] << 16);
        *reinterpret_cast<u32x4t *>(e.dstT + ((int64_t)(c0 + c) * 9 + (8 - tt)) * e.co + o0 + o8) = w;
    }
}

// fp32 -> bf16 copies of many tensors in ONE launch (the 1x1 weights of a segment): records {src, dst, n, block0}, a workgroup
// converts 8192 consecutive elements of its tensor (n a multiple of 8; torch._foreach_copy_ with a dtype change ran at ~1.2 TB/s:
// 20-38 us per segment of a ResNet-101)
struct CastRec {
    const float *src;
    uint16_t *dst;
    int64_t n, block0;
};

__global__ __launch_bounds__(256) void cast_many_bf16_kernel(const CastRec *__restrict__ tab, int n) {
    const int64_t b = blockIdx.x;
    int k = 0;
    for (int i = 1; i < n; ++i) k = tab[i].block0 <= b ? i : k;
    const CastRec e = tab[k];
    const int64_t base = (b - e.block0) * 8192;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t i = base + ((int64_t)u * 256 + threadIdx.x) * 8;
        if (i < e.n) {
            float f[8];
            load8f(e.src + i, f);
            *reinterpret_cast<u32x4t *>(e.dst + i) = pack8(f);
        }
    }
}

// y[b, ho, wo, :] = x[b, 2 ho, 2 wo, :] (the row subsample in front of a stride-2 1x1 convolution), 16 bytes per thread
__global__ __launch_bounds__(256) void subsample2_bf16_kernel(const u32x4t *__restrict__ x, int H, int W, int c8, int Ho, int Wo,
                                                              int64_t n, u32x4t *__restrict__ y) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int cg = (int)(e % c8);
    int64_t q = e / c8;
    const int wo = (int)(q % Wo);
    q /= Wo;
    const int ho = (int)(q % Ho);
    const int64_t b = q / Ho;
    y[e] = x[((b * H + 2 * ho) * W + 2 * wo) * c8 + cg];
}

// its gradient: dx[b, h, w, :] = dy[b, h / 2, w / 2, :] at even (h, w), zero elsewhere -- the whole tensor written once
__global__ __launch_bounds__(256) void upsample2_zero_bf16_kernel(const u32x4t *__restrict__ dy, int H, int W, int c8, int Ho, int Wo,
                                                                  int64_t n, u32x4t *__restrict__ dx) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int cg = (int)(e % c8);
    int64_t q = e / c8;
    const int w = (int)(q % W);
    q /= W;
    const int h = (int)(q % H);
    const int64_t b = q / H;
    u32x4t v = {0u, 0u, 0u, 0u};
    if (!((h | w) & 1)) v = dy[((b * Ho + (h >> 1)) * Wo + (w >> 1)) * c8 + cg];
    dx[e] = v;
}

static inline bool bn_shape_ok(int64_t rows, int C) {
    if (rows <= 0 || C <= 0 || (C & 7)) return false;
    const int c8 = C / 8;
    return c8 <= 256 && 256 % c8 == 0;
}

// statistics kernels: (row groups, channel tiles); row groups x 2C atomics <= 128 k, >= 8 passes per workgroup
static inline dim3 bn_stat_grid(int64_t rows, int c8, int groups) {      // rows: of ONE statistics group
    const int c8t = c8 < 32 ? c8 : 32, rpp = 256 / c8t;
    int64_t g = (rows + (int64_t)rpp * 8 - 1) / ((int64_t)rpp * 8);
    int64_t cap = 65536 / (8 * c8) < 256 ? 65536 / (8 * c8) : 256;
    cap = cap / groups > 0 ? cap / groups : 1;                             // (the atomics of all groups count)
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return dim3((unsigned)g, (unsigned)(c8 / c8t), (unsigned)groups);
}

// elementwise kernels: every thread gets >= `min_iters` rows where the tensor allows, at most `cap` workgroups
static inline unsigned bn_grid(int64_t rows, int c8, int min_iters, int cap) {
    const int rpp = 256 / c8;
    int64_t g = (rows + (int64_t)rpp * min_iters - 1) / ((int64_t)rpp * min_iters);
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

}  // namespace dmm

// rows = ALL rows; `groups` statistics groups of rows / groups consecutive rows each (groups calls of the layer on the groups'
// row ranges, in order, in one launch).  stats / saved / sums: [groups][2][C].
static inline bool bn_groups_ok(int64_t rows, int groups) { return groups >= 1 && groups <= 64 && rows % groups == 0; }

extern "C" int dmm_bn_stats_grouped_bf16(const void *x, int64_t rows, int C, int groups, float *stats, dmm_stream_t stream) {
    if (rows < 0 || C <= 0 || !bn_groups_ok(rows, groups)) return DMM_ERR_BAD_ARG;
    if (rows == 0) return DMM_OK;
    if (!x || !stats) return DMM_ERR_BAD_ARG;
    if (!dmm::bn_shape_ok(rows, C)) return DMM_ERR_UNSUPPORTED;
    const int c8 = C / 8;
    hipLaunchKernelGGL(dmm::bn_stats_bf16_kernel, dmm::bn_stat_grid(rows / groups, c8, groups), dim3(256), 0,
                       (hipStream_t)stream, (const uint16_t *)x, rows / groups, c8, stats);
    return dmm::check_launch();
}

extern "C" int dmm_bn_stats_bf16(const void *x, int64_t rows, int C, float *stats, dmm_stream_t stream) {
    return dmm_bn_stats_grouped_bf16(x, rows, C, 1, stats, stream);
}

extern "C" int dmm_bn_apply_grouped_bf16(const void *x, const void *residual, int64_t rows, int C, int groups,
                                         const float *stats, const float *weight, const float *bias, float *running_mean,
                                         float *running_var, float momentum, float eps, int relu, void *y, float *saved,
                                         dmm_stream_t stream) {
    if (rows < 0 || C <= 0 || !bn_groups_ok(rows, groups)) return DMM_ERR_BAD_ARG;
    if (rows == 0) return DMM_OK;
    if (!x || !stats || !weight || !bias || !y || !saved || (!running_mean) != (!running_var)) return DMM_ERR_BAD_ARG;
    if (!dmm::bn_shape_ok(rows, C)) return DMM_ERR_UNSUPPORTED;
    const int c8 = C / 8;
    const int64_t grows = rows / groups;
    const dim3 grid(dmm::bn_grid(grows, c8, 2, 4096 / groups), 1, (unsigned)groups);
#define DMM_BNA(RES_, RELU_)                                                                                             \
    hipLaunchKernelGGL((dmm::bn_apply_bf16_kernel<RES_, RELU_>), grid, dim3(256), 0, (hipStream_t)stream,                \
                       (const uint16_t *)x, (const uint16_t *)residual, grows, c8, stats, weight, bias, running_mean,    \
                       running_var, momentum, eps, (uint16_t *)y, saved)
    if (residual) { if (relu) DMM_BNA(true, true); else DMM_BNA(true, false); }
    else { if (relu) DMM_BNA(false, true); else DMM_BNA(false, false); }
#undef DMM_BNA
    return dmm::check_launch();
}

extern "C" int dmm_bn_apply_bf16(const void *x, const void *residual, int64_t rows, int C, const float *stats,
                                 const float *weight, const float *bias, float *running_mean, float *running_var,
                                 float momentum, float eps, int relu, void *y, float *saved, dmm_stream_t stream) {
    return dmm_bn_apply_grouped_bf16(x, residual, rows, C, 1, stats, weight, bias, running_mean, running_var, momentum, eps,
                                     relu, y, saved, stream);
}

extern "C" int dmm_bn_bwd_reduce_grouped_bf16(const void *dy, const void *dy2, const void *x, const void *y, int64_t rows, int C, int groups,
                                              const float *saved, const float *weight, const float *bias, int relu,
                                              float *sums, dmm_stream_t stream) {
    if (rows < 0 || C <= 0 || relu < 0 || relu > 2 || !bn_groups_ok(rows, groups)) return DMM_ERR_BAD_ARG;
    if (rows == 0) return DMM_OK;
    if (!dy || !x || !saved || !sums || (relu == 1 && !y) || (relu == 2 && (!weight || !bias))) return DMM_ERR_BAD_ARG;
    if (!dmm::bn_shape_ok(rows, C)) return DMM_ERR_UNSUPPORTED;
    const int c8 = C / 8;
    const int64_t grows = rows / groups;
    const dim3 grid = dmm::bn_stat_grid(grows, c8, groups);
#define DMM_BNR(R_)                                                                                                    \
    hipLaunchKernelGGL((dmm::bn_bwd_reduce_bf16_kernel<R_>), grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t *)dy, \
                       (const uint16_t *)dy2, (const uint16_t *)x, (const uint16_t *)y, grows, c8, saved, weight, bias, sums)
    if (relu == 2) DMM_BNR(2); else if (relu == 1) DMM_BNR(1); else DMM_BNR(0);
#undef DMM_BNR
    return dmm::check_launch();
}

extern "C" int dmm_bn_bwd_reduce_bf16(const void *dy, const void *x, const void *y, int64_t rows, int C, const float *saved,
                                      const float *weight, const float *bias, int relu, float *sums, dmm_stream_t stream) {
    return dmm_bn_bwd_reduce_grouped_bf16(dy, nullptr, x, y, rows, C, 1, saved, weight, bias, relu, sums, stream);
}

extern "C" int dmm_bn_bwd_dx_grouped_bf16(const void *dy, const void *dy2, const void *x, const void *y, int64_t rows, int C, int groups,
                                          const float *saved, const float *weight, const float *bias, const float *sums,
                                          int relu, void *dx, void *dres, float *dweight, float *dbias, dmm_stream_t stream) {
    if (rows < 0 || C <= 0 || relu < 0 || relu > 2 || !bn_groups_ok(rows, groups)) return DMM_ERR_BAD_ARG;
    if (rows == 0) return DMM_OK;
    if (!dy || !x || !saved || !weight || !sums || !dx || !dweight || !dbias || (relu == 1 && !y) || (relu == 2 && !bias))
        return DMM_ERR_BAD_ARG;
    if (relu == 2 && dres) return DMM_ERR_BAD_ARG;        // (a residual in front of the ReLU: the mask needs the output)
    if (!dmm::bn_shape_ok(rows, C)) return DMM_ERR_UNSUPPORTED;
    const int c8 = C / 8;
    const int64_t grows = rows / groups;
    const dim3 grid(dmm::bn_grid(grows, c8, 2, 4096 / groups), 1, (unsigned)groups);
#define DMM_BND(RELU_, DRES_)                                                                                            \
    hipLaunchKernelGGL((dmm::bn_bwd_dx_bf16_kernel<RELU_, DRES_>), grid, dim3(256), 0, (hipStream_t)stream,              \
                       (const uint16_t *)dy, (const uint16_t *)dy2, (const uint16_t *)x, (const uint16_t *)y, grows, c8, saved,  \
                       weight, bias, sums, (uint16_t *)dx, (uint16_t *)dres, dweight, dbias)
    if (relu == 2) DMM_BND(2, false);
    else if (relu == 1) { if (dres) DMM_BND(1, true); else DMM_BND(1, false); }
    else { if (dres) DMM_BND(0, true); else DMM_BND(0, false); }
#undef DMM_BND
    return dmm::check_launch();
}

extern "C" int dmm_bn_bwd_dx_bf16(const void *dy, const void *x, const void *y, int64_t rows, int C, const float *saved,
                                  const float *weight, const float *bias, const float *sums, int relu, void *dx, void *dres,
                                  float *dweight, float *dbias, dmm_stream_t stream) {
    return dmm_bn_bwd_dx_grouped_bf16(dy, nullptr, x, y, rows, C, 1, saved, weight, bias, sums, relu, dx, dres, dweight, dbias, stream);
}

extern "C" int dmm_wprep3x3_bf16(const void *table, int n, int64_t tiles, dmm_stream_t stream) {
    if (n < 0 || tiles < 0 || tiles > 0x7fffffffLL) return DMM_ERR_BAD_ARG;
    if (n == 0 || tiles == 0) return DMM_OK;
    if (!table) return DMM_ERR_BAD_ARG;
    hipLaunchKernelGGL(dmm::wprep3x3_bf16_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream,
                       (const dmm::WPrep *)table, n);
    return dmm::check_launch();
}

extern "C" int dmm_cast_many_bf16(const void *table, int n, int64_t blocks, dmm_stream_t stream) {
    if (n < 0 || blocks < 0 || blocks > 0x7fffffffLL) return DMM_ERR_BAD_ARG;
    if (n == 0 || blocks == 0) return DMM_OK;
    if (!table) return DMM_ERR_BAD_ARG;
    hipLaunchKernelGGL(dmm::cast_many_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const dmm::CastRec *)table, n);
    return dmm::check_launch();
}

extern "C" int dmm_subsample2_bf16(const void *x, int B, int H, int W, int C, void *y, dmm_stream_t stream) {
    if (B < 0 || H <= 0 || W <= 0 || C <= 0) return DMM_ERR_BAD_ARG;
    if (C & 7) return DMM_ERR_UNSUPPORTED;
    if (B == 0) return DMM_OK;
    if (!x || !y) return DMM_ERR_BAD_ARG;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, c8 = C / 8;
    const int64_t n = (int64_t)B * Ho * Wo * c8;
    hipLaunchKernelGGL(dmm::subsample2_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const dmm::u32x4t *)x, H, W, c8, Ho, Wo, n, (dmm::u32x4t *)y);
    return dmm::check_launch();
}

extern "C" int dmm_upsample2_zero_bf16(const void *dy, int B, int H, int W, int C, void *dx, dmm_stream_t stream) {
    if (B < 0 || H <= 0 || W <= 0 || C <= 0) return DMM_ERR_BAD_ARG;
    if (C & 7) return DMM_ERR_UNSUPPORTED;
    if (B == 0) return DMM_OK;
    if (!dy || !dx) return DMM_ERR_BAD_ARG;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, c8 = C / 8;
    const int64_t n = (int64_t)B * H * W * c8;
    hipLaunchKernelGGL(dmm::upsample2_zero_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const dmm::u32x4t *)dy, H, W, c8, Ho, Wo, n, (dmm::u32x4t *)dx);
    return dmm::check_launch();
}
