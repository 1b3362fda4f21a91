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
// workgroup's pixel range, accumulating in fp32; the threads' partial sums meet in LDS (fp32) and leave the workgroup
// as row blockIdx.x of `partials` ([rows][2C] fp32: C sums, then C sums of squares).  No global atomics: the rows are
// added up in fp64 by bn_finalize, in a fixed order.
__global__ __launch_bounds__(kThreads) void bn_stats_kernel(const f16* __restrict__ x, int x_ld, long long pixels, int G,
                                                            long long pix_per_wg, float* __restrict__ partials) {
    __shared__ float part[kThreads * 16];  // [PL][2C]: every thread's 8 sums and 8 sums of squares
    const int C = G * 8;
    const int PL = kThreads / G;  // pixels in flight per step (G <= kThreads)
    const int g = threadIdx.x % G, pl = threadIdx.x / G;
    if (pl < PL) {
        const long long p0 = (long long)blockIdx.x * pix_per_wg;
        const long long p1 = p0 + pix_per_wg < pixels ? p0 + pix_per_wg : pixels;
        float s[8], q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
        const f16* xg = x + g * 8;
        long long p = p0 + pl;
        for (; p + 3LL * PL < p1; p += 4LL * PL) {  // four loads in flight
            f16x8 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f16x8*>(xg + (p + (long long)u * PL) * x_ld);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = (float)v[u][e];
                    s[e] += f;
                    q[e] = __builtin_fmaf(f, f, q[e]);
                }
        }
        for (; p < p1; p += PL) {
            const f16x8 v = *reinterpret_cast<const f16x8*>(xg + p * x_ld);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)v[e];
                s[e] += f;
                q[e] = __builtin_fmaf(f, f, q[e]);
            }
        }
        float* mine = part + pl * 2 * C + g * 8;
        *reinterpret_cast<f32x4*>(mine) = f32x4{s[0], s[1], s[2], s[3]};
        *reinterpret_cast<f32x4*>(mine + 4) = f32x4{s[4], s[5], s[6], s[7]};
        *reinterpret_cast<f32x4*>(mine + C) = f32x4{q[0], q[1], q[2], q[3]};
        *reinterpret_cast<f32x4*>(mine + C + 4) = f32x4{q[4], q[5], q[6], q[7]};
    }
    __syncthreads();
    // fixed summation order: the result does not depend on scheduling
    float* row = partials + (size_t)blockIdx.x * 2 * C;
    for (int i = threadIdx.x; i < 2 * C; i += kThreads) {
        float a0 = 0.f, a1 = 0.f;
        int j = 0;
        for (; j + 1 < PL; j += 2) {
            a0 += part[j * 2 * C + i];
            a1 += part[(j + 1) * 2 * C + i];
        }
        if (j < PL) a0 += part[j * 2 * C + i];
        row[i] = a0 + a1;
    }
}

// One workgroup per 8 channels: thread (ch = t % 8, rg = t / 8) adds rows rg, rg + 32, ... of its channel's two
// columns in fp64, the 32 row groups meet in LDS, and the first 8 threads finish their channel.
__global__ __launch_bounds__(kThreads) void bn_finalize_kernel(const float* __restrict__ partials, int rows, long long count, int C,
                                                               int Cvalid, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, float momentum,
                                                               float* __restrict__ running_mean, float* __restrict__ running_var,
                                                               float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ double red[2][32][8];
    const int ch = threadIdx.x & 7, rg = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + ch;
    double s = 0.0, q = 0.0;
    if (c < C) {
        const float* col = partials + c;
        int r = rg;
        for (; r + 96 < rows; r += 128) {  // four independent loads in flight per column
            const float s0 = col[(size_t)r * 2 * C], s1 = col[(size_t)(r + 32) * 2 * C], s2 = col[(size_t)(r + 64) * 2 * C],
                        s3 = col[(size_t)(r + 96) * 2 * C];
            const float q0 = col[(size_t)r * 2 * C + C], q1 = col[(size_t)(r + 32) * 2 * C + C], q2 = col[(size_t)(r + 64) * 2 * C + C],
                        q3 = col[(size_t)(r + 96) * 2 * C + C];
            s += ((double)s0 + (double)s1) + ((double)s2 + (double)s3);
            q += ((double)q0 + (double)q1) + ((double)q2 + (double)q3);
        }
        for (; r < rows; r += 32) {
            s += (double)col[(size_t)r * 2 * C];
            q += (double)col[(size_t)r * 2 * C + C];
        }
    }
    red[0][rg][ch] = s;
    red[1][rg][ch] = q;
    __syncthreads();
    if (rg != 0 || c >= C) return;
#pragma unroll
    for (int i = 1; i < 32; ++i) {
        s += red[0][i][ch];
        q += red[1][i][ch];
    }
    if (c >= Cvalid) {  // padding channels stay zero
        scale[c] = 0.f;
        shift[c] = 0.f;
        return;
    }
    const double n = (double)count;
    const double mean = s / n;
    double var = q / n - mean * mean;  // biased, what the normalisation uses
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
    const long long step = (long long)gridDim.x * kThreads;
    for (long long i0 = (long long)blockIdx.x * kThreads + threadIdx.x; i0 < total; i0 += 2 * step) {
        f16x8 v[2], r[2];
        long long p[2];
        int c8[2];
        bool on[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {  // both loads issued before either is used
            const long long i = i0 + u * step;
            on[u] = i < total;
            c8[u] = (int)(i % C8);
            p[u] = i / C8;
            if (on[u]) {
                v[u] = *reinterpret_cast<const f16x8*>(x + p[u] * x_ld + c8[u] * 8);
                if (RES) r[u] = *reinterpret_cast<const f16x8*>(res + p[u] * res_ld + c8[u] * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!on[u]) continue;
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = __builtin_fmaf((float)v[u][e], prm[c8[u] * 8 + e], prm[C + c8[u] * 8 + e]);
                if (LEAKY) f = f > 0.f ? f : f * slope;
                if (RES) f += (float)r[u][e];
                o[e] = (f16)f;
            }
            *reinterpret_cast<f16x8*>(y + p[u] * y_ld + c8[u] * 8) = o;
        }
    }
}

}  // namespace

// Rows of partial sums y5_launch_bn_stats writes for this tensor: one per workgroup, each workgroup taking at least 16
// steps of its PL pixels, at most 512 workgroups.
int y5_bn_stats_row_count(long long pixels, int c) {
    if (pixels <= 0 || c <= 0 || c % 8 != 0 || c > kMaxC) return 0;
    const int PL = kThreads / (c / 8);
    long long per_wg = 16LL * PL;
    if ((pixels + per_wg - 1) / per_wg > 512) per_wg = (pixels + 511) / 512;
    return (int)((pixels + per_wg - 1) / per_wg);
}

int y5_launch_bn_stats(const void* x, int x_ld, long long pixels, int c, float* partials, hipStream_t s) {
    Y5_REQUIRE(x && partials, "bn_stats: null pointer");
    Y5_REQUIRE(c > 0 && c % 8 == 0 && c <= kMaxC && x_ld % 8 == 0 && x_ld >= c, "bn_stats: channels %d / pitch %d must be multiples of 8, channels <= %d", c, x_ld, kMaxC);
    Y5_REQUIRE(pixels > 0, "bn_stats: empty tensor");
    Y5_REQUIRE(y5_aligned16(x) && (reinterpret_cast<uintptr_t>(partials) & 3) == 0, "bn_stats: misaligned pointer");
    const int rows = y5_bn_stats_row_count(pixels, c);
    const long long per_wg = (pixels + rows - 1) / rows;
    Y5_REQUIRE((long long)rows * per_wg >= pixels && (long long)(rows - 1) * per_wg < pixels, "bn_stats: internal row split");
    hipLaunchKernelGGL(bn_stats_kernel, dim3((unsigned)rows), dim3(kThreads), 0, s, static_cast<const f16*>(x), x_ld, pixels, c / 8,
                       per_wg, partials);
    Y5_LAUNCH_CHECK("bn_stats_kernel");
    return Y5_OK;
}

int y5_launch_bn_finalize(const float* partials, int rows, long long count, int c, int c_valid, const float* gamma, const float* beta,
                          float eps, float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                          hipStream_t s) {
    Y5_REQUIRE(partials && scale && shift, "bn_finalize: null pointer");
    Y5_REQUIRE(c > 0 && c_valid >= 0 && c_valid <= c, "bn_finalize: %d valid of %d channels", c_valid, c);
    Y5_REQUIRE(count > 0 && rows > 0, "bn_finalize: no values");
    Y5_REQUIRE(eps >= 0.f && momentum >= 0.f && momentum <= 1.f, "bn_finalize: eps %g / momentum %g out of range", eps, momentum);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((c + 7) / 8), dim3(kThreads), 0, s, partials, rows, count, c, c_valid, gamma, beta,
                       eps, momentum, running_mean, running_var, scale, shift);
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
    long long blocks = (total + 2 * kThreads - 1) / (2 * kThreads);
    if (blocks > 256LL * 8) blocks = 256LL * 8;
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
