// HBM-bound data-movement kernels of the conv forward: image packing (with the Focus gather),
// stride-1 max pooling for SPP, 2x upsampling into a concat slice, and pitched channel copies.
// All of them move 16-byte channel vectors (8 fp16) per lane so that a wave touches 1 KiB per
// instruction, and grid-stride over a capped grid.
#include "y5_common.h"

namespace {

constexpr int kThreads = 256;

inline unsigned grid_for(long long work_items) {
    long long blocks = (work_items + kThreads - 1) / kThreads;
    const long long cap = 256LL * 16;  // 256 CUs x 16 resident workgroups is plenty for streaming
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

// One thread per output pixel; writes y_ld/8 vectors.  focus: out channel q*C + c reads
// x[b, c, 2*oy + (q&1), 2*ox + (q>>1)]  (q = 0..3 -> (row,col) parity (0,0),(1,0),(0,1),(1,1)).
__global__ __launch_bounds__(kThreads) void pack_image_kernel(const f16* __restrict__ x, f16* __restrict__ y, int batch,
                                                              int C, int H, int W, int focus, int y_ld) {
    const int Ho = focus ? H / 2 : H, Wo = focus ? W / 2 : W;
    const long long total = (long long)batch * Ho * Wo;
    const int cvalid = focus ? 4 * C : C;
    for (long long p = (long long)blockIdx.x * kThreads + threadIdx.x; p < total; p += (long long)gridDim.x * kThreads) {
        const int ox = (int)(p % Wo);
        const long long t = p / Wo;
        const int oy = (int)(t % Ho);
        const long long b = t / Ho;
        const f16* xb = x + b * (long long)C * H * W;
        f16* yp = y + p * y_ld;
        for (int c0 = 0; c0 < y_ld; c0 += 8) {
            f16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ch = c0 + e;
                f16 val = (f16)0.f;
                if (ch < cvalid) {
                    if (focus) {
                        const int q = ch / C, c = ch - q * C;
                        val = xb[((long long)c * H + (2 * oy + (q & 1))) * W + 2 * ox + (q >> 1)];
                    } else {
                        val = xb[((long long)ch * H + oy) * W + ox];
                    }
                }
                v[e] = val;
            }
            *reinterpret_cast<f16x8*>(yp + c0) = v;
        }
    }
}

// uint8 -> fp16 / 255: img.half() followed by img /= 255.0 (test.py:152-154).  The half division is evaluated in
// fp32 and rounded once, which is what fp16(float(v) / 255.0f) with IEEE division gives; 16 values per thread.
__global__ __launch_bounds__(kThreads) void normalize_u8_kernel(const uint8_t* __restrict__ x, f16* __restrict__ y, long long n) {
    const long long nvec = n / 16;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += (long long)gridDim.x * kThreads) {
        const uint4 v = reinterpret_cast<const uint4*>(x)[i];
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
        f16x8 lo, hi;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            lo[e] = (f16)((float)((w[e >> 2] >> (8 * (e & 3))) & 255u) / 255.0f);
            hi[e] = (f16)((float)((w[2 + (e >> 2)] >> (8 * (e & 3))) & 255u) / 255.0f);
        }
        reinterpret_cast<f16x8*>(y)[2 * i] = lo;
        reinterpret_cast<f16x8*>(y)[2 * i + 1] = hi;
    }
    for (long long i = nvec * 16 + (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads)
        y[i] = (f16)((float)x[i] / 255.0f);
}

// One thread per (pixel, 8-channel vector).
template <typename ET>
__global__ __launch_bounds__(kThreads) void maxpool_kernel(const f16* __restrict__ x, int x_ld, f16* __restrict__ y, int y_ld,
                                                           int batch, int H, int W, int C8, int k) {
    typedef Elem<ET> E;
    typedef typename E::v8 v8;
    const long long total = (long long)batch * H * W * C8;
    const int r = k / 2;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kThreads) {
        const int c8 = (int)(i % C8);
        const long long p = i / C8;
        const int px = (int)(p % W);
        const long long t = p / W;
        const int py = (int)(t % H);
        const long long b = t / H;
        v8 m;
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = (ET)(-__builtin_inff());
        const int y0 = py - r < 0 ? 0 : py - r, y1 = py + r >= H ? H - 1 : py + r;
        const int x0 = px - r < 0 ? 0 : px - r, x1 = px + r >= W ? W - 1 : px + r;
        for (int yy = y0; yy <= y1; ++yy) {
            const f16* row = x + ((b * H + yy) * (long long)W) * x_ld + c8 * 8;
            for (int xx = x0; xx <= x1; ++xx) {
                const v8 v = *reinterpret_cast<const v8*>(row + (long long)xx * x_ld);
                m = E::max8(m, v);
            }
        }
        *reinterpret_cast<v8*>(y + p * y_ld + c8 * 8) = m;
    }
}

// One thread per (output pixel, 8-channel vector).  Bilinear follows ATen's align_corners=False
// rule: src = max(0, 0.5*(dst + 0.5) - 0.5), i1 = min(i0 + 1, in - 1), fp32 blend, one rounding.
template <typename ET>
__global__ __launch_bounds__(kThreads) void upsample2x_kernel(const f16* __restrict__ x, int x_ld, f16* __restrict__ y, int y_ld,
                                                              int batch, int H, int W, int C8, int mode) {
    typedef typename Elem<ET>::v8 v8;
    const int Ho = 2 * H, Wo = 2 * W;
    const long long total = (long long)batch * Ho * Wo * C8;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kThreads) {
        const int c8 = (int)(i % C8);
        const long long p = i / C8;
        const int ox = (int)(p % Wo);
        const long long t = p / Wo;
        const int oy = (int)(t % Ho);
        const long long b = t / Ho;
        const f16* xb = x + (b * H * (long long)W) * x_ld + c8 * 8;
        v8 out;
        if (mode == 0) {
            out = *reinterpret_cast<const v8*>(xb + ((long long)(oy >> 1) * W + (ox >> 1)) * x_ld);
        } else {
            float sy = 0.5f * (oy + 0.5f) - 0.5f; if (sy < 0.f) sy = 0.f;
            float sx = 0.5f * (ox + 0.5f) - 0.5f; if (sx < 0.f) sx = 0.f;
            const int iy0 = (int)sy, ix0 = (int)sx;
            const int iy1 = iy0 + (iy0 < H - 1 ? 1 : 0), ix1 = ix0 + (ix0 < W - 1 ? 1 : 0);
            const float ly1 = sy - iy0, lx1 = sx - ix0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
            const v8 v00 = *reinterpret_cast<const v8*>(xb + ((long long)iy0 * W + ix0) * x_ld);
            const v8 v01 = *reinterpret_cast<const v8*>(xb + ((long long)iy0 * W + ix1) * x_ld);
            const v8 v10 = *reinterpret_cast<const v8*>(xb + ((long long)iy1 * W + ix0) * x_ld);
            const v8 v11 = *reinterpret_cast<const v8*>(xb + ((long long)iy1 * W + ix1) * x_ld);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                out[e] = (ET)(ly0 * (lx0 * (float)v00[e] + lx1 * (float)v01[e]) + ly1 * (lx0 * (float)v10[e] + lx1 * (float)v11[e]));
        }
        *reinterpret_cast<v8*>(y + p * y_ld + c8 * 8) = out;
    }
}

// SPP in one launch: level l (l = 0..levels-1) is the k x k stride-1 max pool applied l+1 times, which equals
// a single pool of size l*(k-1)+k (max composes exactly).  One workgroup owns one image and one 8-channel vector:
// the H x W plane lives in LDS (two buffers), every level is read from LDS and written to LDS + global.
template <typename ET>
__global__ __launch_bounds__(kThreads) void maxpool_cascade_kernel(const f16* __restrict__ x, int x_ld, f16* __restrict__ y, int y_ld,
                                                                   int y_step, int levels, int H, int W, int C8, int k) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    typename Elem<ET>::v8* buf0 = reinterpret_cast<typename Elem<ET>::v8*>(smem_raw);
    typename Elem<ET>::v8* buf1 = buf0 + H * W;
    const int b = blockIdx.x / C8, c8 = blockIdx.x - b * C8;
    const int npix = H * W, r = k / 2;
    const long long pbase = (long long)b * npix;
    for (int p = threadIdx.x; p < npix; p += kThreads) buf0[p] = *reinterpret_cast<const typename Elem<ET>::v8*>(x + (pbase + p) * x_ld + c8 * 8);
    __syncthreads();
    typename Elem<ET>::v8* src = buf0;
    typename Elem<ET>::v8* dst = buf1;
    for (int l = 0; l < levels; ++l) {
        for (int p = threadIdx.x; p < npix; p += kThreads) {
            const int py = p / W, px = p - py * W;
            const int y0 = py - r < 0 ? 0 : py - r, y1 = py + r >= H ? H - 1 : py + r;
            const int x0 = px - r < 0 ? 0 : px - r, x1 = px + r >= W ? W - 1 : px + r;
            typename Elem<ET>::v8 m = src[y0 * W + x0];
            for (int yy = y0; yy <= y1; ++yy)
                for (int xx = x0; xx <= x1; ++xx) m = Elem<ET>::max8(m, src[yy * W + xx]);
            dst[p] = m;
            *reinterpret_cast<typename Elem<ET>::v8*>(y + (pbase + p) * y_ld + (long long)l * y_step + c8 * 8) = m;
        }
        __syncthreads();
        typename Elem<ET>::v8* t = src; src = dst; dst = t;
    }
}

// The same cascade, separable (a k x k max is a row max of column maxes) and VG 8-channel vectors wide: a workgroup
// owns one image and VG adjacent vectors, so that global accesses are VG*16 contiguous bytes per pixel, and every
// level costs 2k LDS reads per output instead of k*k.  Three LDS planes rotate: level input, row maxes, level output.
template <int VG, typename ET>
__global__ __launch_bounds__(kThreads) void maxpool_cascade_sep_kernel(const f16* __restrict__ x, int x_ld, f16* __restrict__ y,
                                                                       int y_ld, int y_step, int levels, int H, int W, int CG, int k) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int npix = H * W, n = npix * VG, r = k / 2;
    typename Elem<ET>::v8* src = reinterpret_cast<typename Elem<ET>::v8*>(smem_raw);
    typename Elem<ET>::v8* tmp = src + n;
    typename Elem<ET>::v8* dst = tmp + n;
    const int b = blockIdx.x / CG, c0 = (blockIdx.x - b * CG) * VG * 8;
    const long long pbase = (long long)b * npix;
    for (int i = threadIdx.x; i < n; i += kThreads) {
        const int p = i / VG, v = i - p * VG;
        src[i] = *reinterpret_cast<const typename Elem<ET>::v8*>(x + (pbase + p) * x_ld + c0 + v * 8);
    }
    __syncthreads();
    for (int l = 0; l < levels; ++l) {
        for (int i = threadIdx.x; i < n; i += kThreads) {  // along the row
            const int p = i / VG, v = i - p * VG;
            const int py = p / W, px = p - py * W;
            const int x0 = px - r < 0 ? 0 : px - r, x1 = px + r >= W ? W - 1 : px + r;
            typename Elem<ET>::v8 m = src[(py * W + x0) * VG + v];
            for (int xx = x0 + 1; xx <= x1; ++xx) m = Elem<ET>::max8(m, src[(py * W + xx) * VG + v]);
            tmp[i] = m;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += kThreads) {  // along the column
            const int p = i / VG, v = i - p * VG;
            const int py = p / W, px = p - py * W;
            const int y0 = py - r < 0 ? 0 : py - r, y1 = py + r >= H ? H - 1 : py + r;
            typename Elem<ET>::v8 m = tmp[(y0 * W + px) * VG + v];
            for (int yy = y0 + 1; yy <= y1; ++yy) m = Elem<ET>::max8(m, tmp[(yy * W + px) * VG + v]);
            dst[i] = m;
            *reinterpret_cast<typename Elem<ET>::v8*>(y + (pbase + p) * y_ld + (long long)l * y_step + c0 + v * 8) = m;
        }
        __syncthreads();
        typename Elem<ET>::v8* t = src; src = dst; dst = t;
    }
}

// The 5 x 5 cascade (SPP's 5 / 9 / 13 pools) for planes of up to MAXD x MAXD pixels, in REGISTERS: a thread owns one row
// (then one column) of one 8-channel vector.  The 9-wide window at x is the union of the 5-wide windows at x - 2 and x + 2,
// the 13-wide one the union of the 9-wide windows at x - 2 and x + 2, so every level costs one max per element after the
// first; positions outside the plane hold -inf (what max pooling pads with).  Row maxes of all levels go through LDS once
// ([level][pixel][vector]), the column pass stores straight to memory.  The LDS traffic is 6 accesses per pixel and
// vector instead of 30.
// Walks a line of MAXD elements once and hands emit(level, x, value) the window maxes in the order they complete, so that
// only a few values per level are live at a time.
template <int MAXD, typename V8, typename MAX, typename EMIT>
__device__ __forceinline__ void window_maxes(const V8 (&a)[MAXD], V8 neg, int levels, MAX mx, EMIT emit) {
    V8 m5[MAXD + 4], m9[MAXD + 2];
    auto e = [&](int i) { return i >= 2 && i < MAXD + 2 ? a[i - 2] : neg; };  // e(i + 2) = element i (-inf outside the line)
#pragma unroll
    for (int j = 0; j < MAXD + 4; ++j) {
        // elements j - 2 .. j + 2 = e(j) .. e(j + 4)
        m5[j] = mx(mx(mx(e(j), e(j + 1)), mx(e(j + 2), e(j + 3))), e(j + 4));
        if (j < MAXD) emit(0, j, m5[j]);
        if (levels > 1 && j >= 2) {
            const int x9 = j - 2;
            m9[x9] = mx(m5[x9 < 2 ? 0 : x9 - 2], m5[j]);
            if (x9 < MAXD) emit(1, x9, m9[x9]);
            if (levels > 2 && x9 >= 2) {
                const int x13 = x9 - 2;
                emit(2, x13, mx(m9[x13 < 2 ? 0 : x13 - 2], m9[x9]));
            }
        }
    }
}

// max of eight 16-bit values; fp16 through v_pk_max_f16 (the compiler's fmax re-rounds both operands first)
__device__ __forceinline__ f16x8 pool_max8(f16x8 p, f16x8 q) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 a = __builtin_bit_cast(u32x4, p), b = __builtin_bit_cast(u32x4, q);
    const u32x4 r = {pk_max2(a[0], b[0]), pk_max2(a[1], b[1]), pk_max2(a[2], b[2]), pk_max2(a[3], b[3])};
    return __builtin_bit_cast(f16x8, r);
}
__device__ __forceinline__ bf16x8 pool_max8(bf16x8 p, bf16x8 q) { return Elem<bf16>::max8(p, q); }

// Threads [0, 128) take the rows (all levels: each level's row maxes come from the one before); then threads
// [128 l, 128 l + 128) take the columns of level l.
template <int VG, int MAXD, typename ET>
__global__ __launch_bounds__(384) void maxpool_cascade_reg_kernel(const f16* __restrict__ x, int x_ld, f16* __restrict__ y, int y_ld, int y_step,
                                                                 int levels, int H, int W, int CG) {
    static_assert(VG * MAXD <= 128, "a line per thread, 128 threads per level");
    typedef typename Elem<ET>::v8 V8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    V8* planes = reinterpret_cast<V8*>(smem_raw);  // [level][H][W][VG]
    const int b = blockIdx.x / CG, c0 = (blockIdx.x - b * CG) * VG * 8;
    const long long pbase = (long long)b * H * W;
    const int grp = threadIdx.x >> 7, t = threadIdx.x & 127, line = t / VG, v = t - line * VG;
    V8 neg;
#pragma unroll
    for (int e = 0; e < 8; ++e) neg[e] = (ET)(-__builtin_inff());
    auto mx = [](V8 p, V8 q) { return pool_max8(p, q); };
    V8 a[MAXD];
    if (grp == 0 && line < H) {  // along row `line`
#pragma unroll
        for (int i = 0; i < MAXD; ++i) a[i] = i < W ? *reinterpret_cast<const V8*>(x + (pbase + line * W + i) * x_ld + c0 + v * 8) : neg;
        window_maxes<MAXD>(a, neg, levels, mx, [&](int l, int i, V8 m) {
            if (i < W) planes[((l * H + line) * W + i) * VG + v] = m;
        });
    }
    __syncthreads();
    if (grp < levels && line < W) {  // along column `line` of level grp
        const int l = grp;
#pragma unroll
        for (int i = 0; i < MAXD; ++i) a[i] = i < H ? planes[((l * H + i) * W + line) * VG + v] : neg;
        window_maxes<MAXD>(a, neg, l + 1, mx, [&](int ll, int i, V8 m) {
            if (ll == l && i < H) *reinterpret_cast<V8*>(y + (pbase + i * W + line) * y_ld + (long long)l * y_step + c0 + v * 8) = m;
        });
    }
}

__global__ __launch_bounds__(kThreads) void copy_channels_kernel(const f16* __restrict__ x, int x_ld, f16* __restrict__ y, int y_ld,
                                                                 long long pixels, int C8) {
    const long long total = pixels * C8;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long long)gridDim.x * kThreads) {
        const int c8 = (int)(i % C8);
        const long long p = i / C8;
        *reinterpret_cast<f16x8*>(y + p * y_ld + c8 * 8) = *reinterpret_cast<const f16x8*>(x + p * x_ld + c8 * 8);
    }
}

template <int VG, typename ET>
int launch_pool_reg(const f16* xp, int x_ld, f16* yp, int y_ld, int y_step, int levels, int batch, int h, int w, int c, hipStream_t s) {
    const size_t lds = (size_t)levels * h * w * VG * 16;
    static std::atomic<unsigned long long> attr_set{0};
    const unsigned long long bit = 1ull << y5_current_device();
    if (!(attr_set.load() & bit)) {
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(maxpool_cascade_reg_kernel<VG, 20, ET>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 20 * 20 * VG * 16));
        attr_set.fetch_or(bit);
    }
    hipLaunchKernelGGL((maxpool_cascade_reg_kernel<VG, 20, ET>), dim3((unsigned)(batch * (c / (8 * VG)))), dim3(384), lds, s, xp, x_ld, yp, y_ld, y_step, levels, h, w, c / (8 * VG));
    Y5_LAUNCH_CHECK("maxpool_cascade_reg_kernel");
    return Y5_OK;
}

}  // namespace

int y5_launch_pack_image(const void* x, void* y, int batch, int c, int h, int w, int focus, int y_ld, hipStream_t s) {
    Y5_REQUIRE(x && y, "pack_image: null pointer");
    Y5_REQUIRE(batch > 0 && c > 0 && h > 0 && w > 0, "pack_image: empty tensor");
    Y5_REQUIRE(!focus || (h % 2 == 0 && w % 2 == 0), "pack_image: focus needs even h,w (got %dx%d)", h, w);
    Y5_REQUIRE(y_ld % 8 == 0 && y_ld >= (focus ? 4 * c : c), "pack_image: y_ld %d too small or not a multiple of 8", y_ld);
    Y5_REQUIRE(y5_aligned16(y), "pack_image: output must be 16-byte aligned");
    const long long total = (long long)batch * (focus ? h / 2 : h) * (focus ? w / 2 : w);
    hipLaunchKernelGGL(pack_image_kernel, dim3(grid_for(total)), dim3(kThreads), 0, s, static_cast<const f16*>(x),
                       static_cast<f16*>(y), batch, c, h, w, focus, y_ld);
    Y5_LAUNCH_CHECK("pack_image_kernel");
    return Y5_OK;
}

int y5_launch_normalize_u8(const void* x, void* y, long long n, hipStream_t s) {
    Y5_REQUIRE(n >= 0 && (n == 0 || (x && y)), "normalize_u8: null pointer");
    if (n == 0) return Y5_OK;
    Y5_REQUIRE(y5_aligned16(x) && y5_aligned16(y), "normalize_u8: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(normalize_u8_kernel, dim3(grid_for((n + 15) / 16)), dim3(kThreads), 0, s, static_cast<const uint8_t*>(x),
                       static_cast<f16*>(y), n);
    Y5_LAUNCH_CHECK("normalize_u8_kernel");
    return Y5_OK;
}

int y5_launch_maxpool(const void* x, int x_ld, void* y, int y_ld, int batch, int h, int w, int c, int k, hipStream_t s, int dtype) {
    Y5_REQUIRE(dtype == Y5_F16 || dtype == Y5_BF16, "maxpool: dtype %d", dtype);
    Y5_REQUIRE(x && y, "maxpool: null pointer");
    Y5_REQUIRE(k % 2 == 1 && k >= 1 && k <= 13, "maxpool: k %d must be odd and <= 13", k);
    Y5_REQUIRE(c > 0 && c % 8 == 0 && x_ld % 8 == 0 && y_ld % 8 == 0 && x_ld >= c && y_ld >= c, "maxpool: channels/pitches must be multiples of 8");
    Y5_REQUIRE(batch > 0 && h > 0 && w > 0, "maxpool: empty tensor");
    Y5_REQUIRE(y5_aligned16(x) && y5_aligned16(y), "maxpool: pointers must be 16-byte aligned");
    const long long total = (long long)batch * h * w * (c / 8);
    if (dtype == Y5_BF16)
        hipLaunchKernelGGL(maxpool_kernel<bf16>, dim3(grid_for(total)), dim3(kThreads), 0, s, static_cast<const f16*>(x), x_ld,
                           static_cast<f16*>(y), y_ld, batch, h, w, c / 8, k);
    else
        hipLaunchKernelGGL(maxpool_kernel<f16>, dim3(grid_for(total)), dim3(kThreads), 0, s, static_cast<const f16*>(x), x_ld,
                           static_cast<f16*>(y), y_ld, batch, h, w, c / 8, k);
    Y5_LAUNCH_CHECK("maxpool_kernel");
    return Y5_OK;
}

int y5_launch_upsample2x(const void* x, int x_ld, void* y, int y_ld, int batch, int h, int w, int c, int mode,
                         hipStream_t s, int dtype) {
    Y5_REQUIRE(dtype == Y5_F16 || dtype == Y5_BF16, "upsample2x: dtype %d", dtype);
    Y5_REQUIRE(x && y, "upsample2x: null pointer");
    Y5_REQUIRE(mode == 0 || mode == 1, "upsample2x: mode %d not in {0 nearest, 1 bilinear}", mode);
    Y5_REQUIRE(c > 0 && c % 8 == 0 && x_ld % 8 == 0 && y_ld % 8 == 0 && x_ld >= c && y_ld >= c, "upsample2x: channels/pitches must be multiples of 8");
    Y5_REQUIRE(batch > 0 && h > 0 && w > 0, "upsample2x: empty tensor");
    Y5_REQUIRE(y5_aligned16(x) && y5_aligned16(y), "upsample2x: pointers must be 16-byte aligned");
    const long long total = (long long)batch * 4 * h * w * (c / 8);
    if (dtype == Y5_BF16)
        hipLaunchKernelGGL(upsample2x_kernel<bf16>, dim3(grid_for(total)), dim3(kThreads), 0, s, static_cast<const f16*>(x), x_ld,
                           static_cast<f16*>(y), y_ld, batch, h, w, c / 8, mode);
    else
        hipLaunchKernelGGL(upsample2x_kernel<f16>, dim3(grid_for(total)), dim3(kThreads), 0, s, static_cast<const f16*>(x), x_ld,
                           static_cast<f16*>(y), y_ld, batch, h, w, c / 8, mode);
    Y5_LAUNCH_CHECK("upsample2x_kernel");
    return Y5_OK;
}

int y5_launch_copy_channels(const void* x, int x_ld, void* y, int y_ld, int64_t pixels, int c, hipStream_t s) {
    Y5_REQUIRE(x && y, "copy_channels: null pointer");
    Y5_REQUIRE(c > 0 && c % 8 == 0 && x_ld % 8 == 0 && y_ld % 8 == 0 && x_ld >= c && y_ld >= c, "copy_channels: channels/pitches must be multiples of 8");
    Y5_REQUIRE(pixels > 0, "copy_channels: empty tensor");
    Y5_REQUIRE(y5_aligned16(x) && y5_aligned16(y), "copy_channels: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(copy_channels_kernel, dim3(grid_for(pixels * (c / 8))), dim3(kThreads), 0, s,
                       static_cast<const f16*>(x), x_ld, static_cast<f16*>(y), y_ld, (long long)pixels, c / 8);
    Y5_LAUNCH_CHECK("copy_channels_kernel");
    return Y5_OK;
}

int y5_launch_maxpool_cascade(const void* x, int x_ld, void* y, int y_ld, int y_step, int levels, int batch, int h, int w, int c,
                              int k, hipStream_t s, int dtype) {
    Y5_REQUIRE(dtype == Y5_F16 || dtype == Y5_BF16, "maxpool_cascade: dtype %d", dtype);
    const bool bf = dtype == Y5_BF16;
    Y5_REQUIRE(x && y, "maxpool_cascade: null pointer");
    Y5_REQUIRE(k % 2 == 1 && k >= 1 && k <= 13, "maxpool_cascade: k %d must be odd and <= 13", k);
    Y5_REQUIRE(levels >= 1 && levels <= 8, "maxpool_cascade: levels %d out of range 1..8", levels);
    Y5_REQUIRE(c > 0 && c % 8 == 0 && x_ld % 8 == 0 && y_ld % 8 == 0 && y_step % 8 == 0 && x_ld >= c && y_step >= c, "maxpool_cascade: channels/pitches must be multiples of 8");
    Y5_REQUIRE((long long)(levels - 1) * y_step + c <= y_ld, "maxpool_cascade: levels do not fit the output pitch");
    Y5_REQUIRE(batch > 0 && h > 0 && w > 0, "maxpool_cascade: empty tensor");
    Y5_REQUIRE(y5_aligned16(x) && y5_aligned16(y), "maxpool_cascade: pointers must be 16-byte aligned");
    const f16* xp = static_cast<const f16*>(x);
    f16* yp = static_cast<f16*>(y);
    if (k == 5 && levels <= 3 && h <= 20 && w <= 20 && c % 32 == 0)  // SPP on the coarsest level of an image of up to 640 x 640
        return bf ? launch_pool_reg<4, bf16>(xp, x_ld, yp, y_ld, y_step, levels, batch, h, w, c, s) : launch_pool_reg<4, f16>(xp, x_ld, yp, y_ld, y_step, levels, batch, h, w, c, s);
    const size_t sep = (size_t)3 * h * w * 16;  // separable kernel: three planes per vector
    if (c % 16 == 0 && 2 * sep <= 64 * 1024) {
        if (bf)
            hipLaunchKernelGGL((maxpool_cascade_sep_kernel<2, bf16>), dim3((unsigned)(batch * (c / 16))), dim3(kThreads), 2 * sep, s, xp, x_ld, yp,
                               y_ld, y_step, levels, h, w, c / 16, k);
        else
            hipLaunchKernelGGL((maxpool_cascade_sep_kernel<2, f16>), dim3((unsigned)(batch * (c / 16))), dim3(kThreads), 2 * sep, s, xp, x_ld, yp,
                               y_ld, y_step, levels, h, w, c / 16, k);
        Y5_LAUNCH_CHECK("maxpool_cascade_sep_kernel");
        return Y5_OK;
    }
    if (sep <= 64 * 1024) {
        if (bf)
            hipLaunchKernelGGL((maxpool_cascade_sep_kernel<1, bf16>), dim3((unsigned)(batch * (c / 8))), dim3(kThreads), sep, s, xp, x_ld, yp, y_ld,
                               y_step, levels, h, w, c / 8, k);
        else
            hipLaunchKernelGGL((maxpool_cascade_sep_kernel<1, f16>), dim3((unsigned)(batch * (c / 8))), dim3(kThreads), sep, s, xp, x_ld, yp, y_ld,
                               y_step, levels, h, w, c / 8, k);
        Y5_LAUNCH_CHECK("maxpool_cascade_sep_kernel");
        return Y5_OK;
    }
    const size_t lds = (size_t)2 * h * w * 16;
    Y5_REQUIRE(lds <= 64 * 1024, "maxpool_cascade: %dx%d plane does not fit in LDS", h, w);
    if (bf)
        hipLaunchKernelGGL(maxpool_cascade_kernel<bf16>, dim3((unsigned)(batch * (c / 8))), dim3(kThreads), lds, s, static_cast<const f16*>(x),
                           x_ld, static_cast<f16*>(y), y_ld, y_step, levels, h, w, c / 8, k);
    else
        hipLaunchKernelGGL(maxpool_cascade_kernel<f16>, dim3((unsigned)(batch * (c / 8))), dim3(kThreads), lds, s, static_cast<const f16*>(x),
                           x_ld, static_cast<f16*>(y), y_ld, y_step, levels, h, w, c / 8, k);
    Y5_LAUNCH_CHECK("maxpool_cascade_kernel");
    return Y5_OK;
}
