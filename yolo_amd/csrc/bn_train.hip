// Train-mode BatchNorm2d over NHWC fp16 tensors, the arithmetic of the statistics-recalibration forward
// (filter_pruning/fp_train.py:231-240 runs the model in train() mode under no_grad, so that every
// nn.BatchNorm2d normalises with the batch's own mean / biased variance and moves its running statistics,
// models/common.py:27 and :62).  Three HBM-bound kernels:
//   bn_stats     per-channel sum and sum of squares of a convolution's raw output       (reads the tensor once)
//   bn_finalize  sums -> (scale, shift) of the normalisation + running-statistics update (one thread per channel)
//   affine_act   y = act(x * scale[c] + shift[c]) (+ residual)                          (reads once, writes once)
#include "y5_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxC = 2048;

// Thread t handles channel group t % G (8 channels, one 16-byte load) of pixels t / G, t / G + PL, ... of the
// workgroup's pixel range, accumulating in fp32; partial sums meet in LDS (fp32) and leave the workgroup as one
// fp64 atomic per channel and moment, so that the cross-workgroup order does not show in the result.
__global__ __launch_bounds__(kThreads) void bn_stats_kernel(const f16* __restrict__ x, int x_ld, long long pixels, int G,
                                                            long long pix_per_wg, double* __restrict__ sums) {
    __shared__ float part[2 * kMaxC];
    const int C = G * 8;
    for (int i = threadIdx.x; i < 2 * C; i += kThreads) part[i] = 0.f;
    __syncthreads();
    const int PL = kThreads / G;  // pixels in flight per step (G <= kThreads)
    const int g = threadIdx.x % G, pl = threadIdx.x / G;
    if (pl < PL) {
        const long long p0 = (long long)blockIdx.x * pix_per_wg;
        const long long p1 = p0 + pix_per_wg < pixels ? p0 + pix_per_wg : pixels;
        float s[8], q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
        for (long long p = p0 + pl; p < p1; p += PL) {
            const f16x8 v = *reinterpret_cast<const f16x8*>(x + p * x_ld + g * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)v[e];
                s[e] += f;
                q[e] = __builtin_fmaf(f, f, q[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            atomicAdd(&part[g * 8 + e], s[e]);
            atomicAdd(&part[C + g * 8 + e], q[e]);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += kThreads) atomicAdd(&sums[i], (double)part[i]);
}

__global__ __launch_bounds__(kThreads) void bn_finalize_kernel(const double* __restrict__ sums, long long count, int C, int Cvalid,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float eps, float momentum, float* __restrict__ running_mean,
                                                               float* __restrict__ running_var, float* __restrict__ scale,
                                                               float* __restrict__ shift) {
    const int c = blockIdx.x * kThreads + threadIdx.x;
    if (c >= C) return;
    if (c >= Cvalid) {  // padding channels stay zero
        scale[c] = 0.f;
        shift[c] = 0.f;
        return;
    }
    const double n = (double)count;
    const double mean = sums[c] / n;
    double var = sums[C + c] / n - mean * mean;  // biased, what the normalisation uses
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const double ga = gamma ? (double)gamma[c] : 1.0, be = beta ? (double)beta[c] : 0.0;
    scale[c] = (float)(ga * invstd);
    shift[c] = (float)(be - mean * ga * invstd);
    if (running_mean) running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
    if (running_var) {
        const double unbiased = count > 1 ? var * n / (n - 1.0) : var;
        running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
    }
}

template <bool RES, bool LEAKY>
__global__ __launch_bounds__(kThreads) void affine_act_kernel(const f16* __restrict__ x, int x_ld, f16* __restrict__ y, int y_ld,
                                                              const f16* __restrict__ res, int res_ld, long long pixels, int C8,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              float slope) {
    __shared__ float prm[2 * kMaxC];
    const int C = C8 * 8;
    for (int i = threadIdx.x; i < C; i += kThreads) {
        prm[i] = scale[i];
        prm[C + i] = shift[i];
    }
    __syncthreads();
    const long long total = pixels * C8;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kThreads) {
        const int c8 = (int)(i % C8);
        const long long p = i / C8;
        const f16x8 v = *reinterpret_cast<const f16x8*>(x + p * x_ld + c8 * 8);
        f16x8 r;
        if (RES) r = *reinterpret_cast<const f16x8*>(res + p * res_ld + c8 * 8);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = __builtin_fmaf((float)v[e], prm[c8 * 8 + e], prm[C + c8 * 8 + e]);
            if (LEAKY) f = f > 0.f ? f : f * slope;
            if (RES) f += (float)r[e];
            o[e] = (f16)f;
        }
        *reinterpret_cast<f16x8*>(y + p * y_ld + c8 * 8) = o;
    }
}

}  // namespace

int y5_launch_bn_stats(const void* x, int x_ld, long long pixels, int c, double* sums, hipStream_t s) {
    Y5_REQUIRE(x && sums, "bn_stats: null pointer");
    Y5_REQUIRE(c > 0 && c % 8 == 0 && c <= kMaxC && x_ld % 8 == 0 && x_ld >= c, "bn_stats: channels %d / pitch %d must be multiples of 8, channels <= %d", c, x_ld, kMaxC);
    Y5_REQUIRE(pixels > 0, "bn_stats: empty tensor");
    Y5_REQUIRE(y5_aligned16(x) && (reinterpret_cast<uintptr_t>(sums) & 7) == 0, "bn_stats: misaligned pointer");
    Y5_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 2 * c, s));
    const int G = c / 8;
    // per-thread fp32 partials stay short (<= 64 values) on large tensors; at most 2048 workgroups
    const int PL = kThreads / G;  // G <= kMaxC / 8 = kThreads
    long long pix_per_wg = 64LL * PL;
    long long grid = (pixels + pix_per_wg - 1) / pix_per_wg;
    if (grid > 2048) {
        grid = 2048;
        pix_per_wg = (pixels + grid - 1) / grid;
        grid = (pixels + pix_per_wg - 1) / pix_per_wg;
    }
    hipLaunchKernelGGL(bn_stats_kernel, dim3((unsigned)grid), dim3(kThreads), 0, s, static_cast<const f16*>(x), x_ld, pixels, G,
                       pix_per_wg, sums);
    Y5_LAUNCH_CHECK("bn_stats_kernel");
    return Y5_OK;
}

int y5_launch_bn_finalize(const double* sums, long long count, int c, int c_valid, const float* gamma, const float* beta, float eps,
                          float momentum, float* running_mean, float* running_var, float* scale, float* shift, hipStream_t s) {
    Y5_REQUIRE(sums && scale && shift, "bn_finalize: null pointer");
    Y5_REQUIRE(c > 0 && c_valid >= 0 && c_valid <= c, "bn_finalize: %d valid of %d channels", c_valid, c);
    Y5_REQUIRE(count > 0, "bn_finalize: no values");
    Y5_REQUIRE(eps >= 0.f && momentum >= 0.f && momentum <= 1.f, "bn_finalize: eps %g / momentum %g out of range", eps, momentum);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((c + kThreads - 1) / kThreads), dim3(kThreads), 0, s, sums, count, c, c_valid, gamma,
                       beta, eps, momentum, running_mean, running_var, scale, shift);
    Y5_LAUNCH_CHECK("bn_finalize_kernel");
    return Y5_OK;
}

int y5_launch_affine_act(const void* x, int x_ld, void* y, int y_ld, const void* res, int res_ld, long long pixels, int c,
                         const float* scale, const float* shift, int act, float slope, hipStream_t s) {
    Y5_REQUIRE(x && y && scale && shift, "affine_act: null pointer");
    Y5_REQUIRE(c > 0 && c % 8 == 0 && c <= kMaxC && x_ld % 8 == 0 && y_ld % 8 == 0 && x_ld >= c && y_ld >= c, "affine_act: channels/pitches must be multiples of 8, channels <= %d", kMaxC);
    Y5_REQUIRE(!res || (res_ld % 8 == 0 && res_ld >= c), "affine_act: residual pitch %d", res_ld);
    Y5_REQUIRE(pixels > 0, "affine_act: empty tensor");
    Y5_REQUIRE(act == Y5_ACT_NONE || act == Y5_ACT_LEAKY, "affine_act: activation %d", act);
    Y5_REQUIRE(y5_aligned16(x) && y5_aligned16(y) && y5_aligned16(res), "affine_act: pointers must be 16-byte aligned");
    const long long total = pixels * (c / 8);
    long long blocks = (total + kThreads - 1) / kThreads;
    if (blocks > 256LL * 16) blocks = 256LL * 16;
    const dim3 grid((unsigned)blocks), block(kThreads);
    const f16 *xp = static_cast<const f16*>(x), *rp = static_cast<const f16*>(res);
    f16* yp = static_cast<f16*>(y);
    const bool leaky = act == Y5_ACT_LEAKY;
    if (res && leaky) hipLaunchKernelGGL((affine_act_kernel<true, true>), grid, block, 0, s, xp, x_ld, yp, y_ld, rp, res_ld, pixels, c / 8, scale, shift, slope);
    else if (res) hipLaunchKernelGGL((affine_act_kernel<true, false>), grid, block, 0, s, xp, x_ld, yp, y_ld, rp, res_ld, pixels, c / 8, scale, shift, slope);
    else if (leaky) hipLaunchKernelGGL((affine_act_kernel<false, true>), grid, block, 0, s, xp, x_ld, yp, y_ld, rp, res_ld, pixels, c / 8, scale, shift, slope);
    else hipLaunchKernelGGL((affine_act_kernel<false, false>), grid, block, 0, s, xp, x_ld, yp, y_ld, rp, res_ld, pixels, c / 8, scale, shift, slope);
    Y5_LAUNCH_CHECK("affine_act_kernel");
    return Y5_OK;
}
