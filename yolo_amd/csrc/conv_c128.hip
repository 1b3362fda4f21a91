// 3x3 stride-1 convolution 128 -> 128 (+ residual) followed by a 1x1 convolution 128 -> 128, as one kernel with ALL 3x3
// weights in registers: the Bottleneck.cv2 of a 128-channel BottleneckCSP chained to the next Bottleneck's cv1 or to the
// block's cv3 (models/common.py:43, :61; the 40x40 stages of yolov5s).  Same contract as the other y5_conv_chain_f16
// variants.
//
// The tiled kernels stream these layers' 288 KiB of 3x3 weights (and, as im2col, every input pixel nine times) from L2
// into LDS once per 128-pixel tile: 576 KiB per tile against 8 k cycles of matrix work per CU, which is more than the
// L2s deliver to 256 CUs at once.  Here a workgroup owns a strip of five full-width image rows (200 pixels at 40 x 40: one
// strip per CU at batch 32) and nothing but its pixels moves after the start:
//  * eight waves of 256 registers hold the whole weight tensor between them - wave (cb, kh) keeps the rows of output
//    channels [32 cb, +32) for input channels [64 kh, +64) of all nine taps (36 fragments = 144 registers);
//  * the strip with its one-pixel halo (7 x 42 pixels, [pixel][channel] rows, 16-byte slot s of pixel p at s ^ (p & 15);
//    the halo columns and out-of-image rows are zeros supplied by the copies' range check) is copied global -> LDS once;
//  * the strip is walked in passes of 64 pixels: every wave multiplies its channel block's half-K partial sums for both
//    32-pixel groups straight from the staged strip, the kh = 1 waves hand theirs over through LDS, the kh = 0 waves add,
//    apply the post-ops (+ residual) and leave the fp16 result tile in LDS, and all eight waves multiply that tile with
//    the 1x1 weights (staged once in LDS) and store.
#include <atomic>

#include "y5_common.h"

namespace {

#include "conv_common.h"
#include "epi_mix.h"

struct C128Args {
    const f16 *x, *res;
    f16 *y1, *y2;
    const f16 *w1, *w2;
    const float *b1, *s1, *h1, *b2, *s2, *h2;
    int w1_ld, w2_ld;
    int H, W, x_ld, res_ld, y1_ld, y2_ld, x_bytes;
    int strips_per_image;
    FastDiv div_w, div_spi;
    float slope1, slope2;
};

constexpr int C = 128, R = 5, PW = 42, HPX = (R + 2) * PW;  // rows per strip; staged row pitch and pixels (strip + halo, width <= 40)
constexpr int ROW = C * 2;                                   // bytes per staged pixel: 16 slots
constexpr int NW = 8, NTH = 64 * NW;
constexpr int X_PIECES = (HPX * ROW + 1023) / 1024, X_BYTES = X_PIECES * 1024, MYP = (X_PIECES + NW - 1) / NW;
constexpr int PASS = 64;                                     // pixels per pass: two 32-pixel groups
constexpr int RED0 = X_BYTES, RED_BYTES = 4 * 2 * 64 * 64;   // partial sums of the kh = 1 waves: [cb][group][4][lane] x 16 bytes
constexpr int V0 = RED0 + RED_BYTES, V_BYTES = PASS * ROW;   // the first result of a pass
constexpr int W20 = V0 + V_BYTES, W2_PIECES = C * ROW / 1024;  // 1x1 weights: row i (slot s at s ^ (i & 15)) = output channel (i & ~31) + perm(i & 31)
constexpr int PRM0 = W20 + C * ROW, SMEM = PRM0 + 2 * 3 * C * 4;
static_assert(RED_BYTES >= NW * 4096 && RED0 % 1024 == 0 && W20 % 1024 == 0 && W2_PIECES % NW == 0, "DMA pieces land on 1 KiB boundaries");
static_assert(SMEM <= 160 * 1024, "fits the CU's LDS");

#ifdef Y5_STAMPS
// Diagnostic build only: per wave, cycles in {weights + parameters requested, copies issued, everything landed, first barrier,
// products, barrier, sum + post-ops, barrier, 1x1 + stores} (the last five summed over the passes)
__device__ unsigned long long y5_c128_stamp_buf[256 * 8 * 12];
#define C128_STAMP(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[i] += now_ - st_last; st_last = now_; }
#define C128_STAMP_ARGS , unsigned long long (&st_acc)[12], unsigned long long& st_last
#define C128_STAMP_PASS , st_acc, st_last
#else
#define C128_STAMP(i)
#define C128_STAMP_ARGS
#define C128_STAMP_PASS
#endif

template <int NPT, int F1, int F2>  // 32-pixel groups in this pass; post-ops of the two convolutions (bit 0 bias, bit 1 scale + shift)
__device__ __forceinline__ void pass(const C128Args& a, unsigned char* smem, const f16x8 (&wf)[36], int p0, int npx, long long pix0,
                                     int lr, int lh, int lane, int cb, int kh, int wave, int keep_first, float negzero C128_STAMP_ARGS) {
    const float* prm = reinterpret_cast<const float*>(smem + PRM0);
    // Both waves of a channel block multiply both pixel groups over their half of K; wave (cb, kh) then completes group kh:
    // it hands its partial sums of the other group over through LDS and adds the partner's partial sums of its own.  (A pass
    // of one group is completed by the kh = 0 waves.)
    const bool mine = kh < NPT;
    int hp0[NPT];
    f16x8 rl, rh;
#pragma unroll
    for (int nt = 0; nt < NPT; ++nt) {
        const int p = p0 + nt * 32 + lr, pv = p < npx ? p : npx - 1;
        const int row = (int)fdiv((unsigned)pv, a.div_w);
        hp0[nt] = row * PW + pv - row * a.W;
    }
    const int pm = p0 + kh * 32 + lr;  // the pixel this lane completes
    if (mine && a.res && pm < npx) {   // its 16 residual values, fetched under the products
        const f16* rp = a.res + (size_t)(pix0 + pm) * a.res_ld + cb * 32 + 16 * lh;
        rl = *reinterpret_cast<const f16x8*>(rp);
        rh = *reinterpret_cast<const f16x8*>(rp + 8);
    }
    f32x16 acc[NPT];
#pragma unroll
    for (int nt = 0; nt < NPT; ++nt) acc[nt] = zero16();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        int off[NPT];
#pragma unroll
        for (int nt = 0; nt < NPT; ++nt) {
            const int hp = hp0[nt] + (tap / 3) * PW + tap % 3;
            off[nt] = hp * ROW + (((8 * kh + lh) ^ (hp & 15)) << 4);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int nt = 0; nt < NPT; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[tap * 4 + ks], *reinterpret_cast<const f16x8*>(smem + (off[nt] ^ (ks << 5))), acc[nt], 0, 0, 0);
    }
    C128_STAMP(4)
    unsigned char* give = smem + RED0 + (((cb * 2 + kh) * 4 * 64 + lane) << 4);        // this wave's hand-over: [4][lane] x 16 bytes
    const unsigned char* take = smem + RED0 + (((cb * 2 + (kh ^ 1)) * 4 * 64 + lane) << 4);
    if (NPT == 2 || kh == 1) {
        const f32x16& o = acc[NPT == 2 ? (kh ^ 1) : 0];
#pragma unroll
        for (int e = 0; e < 4; ++e) *reinterpret_cast<f32x4*>(give + e * 1024) = (f32x4){o[4 * e], o[4 * e + 1], o[4 * e + 2], o[4 * e + 3]};
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the partial sums are there; the previous pass's 1x1 has read V
    C128_STAMP(5)
    if (mine) {
        f32x16 sum = acc[NPT == 2 ? kh : 0];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x4 o = *reinterpret_cast<const f32x4*>(take + e * 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i) sum[4 * e + i] += o[i];
        }
        f16x8 lo, hi;
        finish<F1>(sum, prm, C, cb * 32 + 16 * lh, a.slope1, negzero, lo, hi);
        const int q = kh * 32 + lr;
        if (a.res && pm < npx) {
            lo += rl;
            hi += rh;
        }
        unsigned char* vp = smem + V0 + q * ROW;
        *reinterpret_cast<f16x8*>(vp + (((4 * cb + 2 * lh) ^ (q & 15)) << 4)) = lo;
        *reinterpret_cast<f16x8*>(vp + (((4 * cb + 2 * lh + 1) ^ (q & 15)) << 4)) = hi;
        if (keep_first && pm < npx) {
            f16* yp = a.y1 + (size_t)(pix0 + pm) * a.y1_ld + cb * 32 + 16 * lh;
            *reinterpret_cast<f16x8*>(yp) = lo;
            *reinterpret_cast<f16x8*>(yp + 8) = hi;
        }
    }
    C128_STAMP(6)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the first result of this pass is complete
    C128_STAMP(7)
    const int cb2 = wave & 3, ph = wave >> 2;  // 1x1: output channels [32 cb2, +32) of pixel group ph
    if (ph < NPT) {
        const int q = ph * 32 + lr;
        const int wrow = cb2 * 32 + lr;
        const unsigned char* wp = smem + W20 + wrow * ROW;
        f32x16 acc2 = zero16();
#pragma unroll
        for (int ks = 0; ks < C / 16; ++ks)
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(wp + (((2 * ks + lh) ^ (wrow & 15)) << 4)),
                                                          *reinterpret_cast<const f16x8*>(smem + V0 + q * ROW + (((2 * ks + lh) ^ (q & 15)) << 4)), acc2, 0, 0, 0);
        f16x8 lo, hi;
        finish<F2>(acc2, prm + 3 * C, C, cb2 * 32 + 16 * lh, a.slope2, negzero, lo, hi);
        const int p = p0 + q;
        if (p < npx) {
            f16* yp = a.y2 + (size_t)(pix0 + p) * a.y2_ld + cb2 * 32 + 16 * lh;
            *reinterpret_cast<f16x8*>(yp) = lo;
            *reinterpret_cast<f16x8*>(yp + 8) = hi;
        }
    }
    C128_STAMP(8)
}

template <int F1, int F2>
__global__ __launch_bounds__(NTH) void conv_c128_kernel(const C128Args a, const int keep_first) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, lr = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int strip = xcd_remap(blockIdx.x, gridDim.x);  // neighbouring strips share halo rows: keep them on one L2
    const int b = (int)fdiv((unsigned)strip, a.div_spi), y0 = (strip - b * a.strips_per_image) * R;
    const int rows = a.H - y0 < R ? a.H - y0 : R, npx = rows * a.W;
    const long long pix0 = ((long long)b * a.H + y0) * a.W;  // the strip's pixels are contiguous in the image

#ifdef Y5_STAMPS
    unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#endif
    // Everything the workgroup needs is requested before anything is waited for, in the order in which it is consumed
    // LAST first: the copies global -> LDS of the strip + halo and of the 1x1 weights, the per-channel parameters, then the
    // 3x3 weights into registers tap by tap.  Vector memory operations retire in order, so the permutation of tap 0 below
    // can start when that tap's four loads have landed, while the later taps are still streaming in.
    const int cb = wave & 3, kh = wave >> 2;
    const i32x4 xr = make_buf(a.x, a.x_bytes);
#pragma unroll
    for (int j = 0; j < MYP; ++j) {
        const int d = wave + NW * j;
        if (d < X_PIECES) {
            const int hp = d * 4 + (lane >> 4), r = hp / PW, c = hp - r * PW, s = (lane & 15) ^ (hp & 15);
            const int iy = y0 - 1 + r;
            const bool ok = hp < HPX && c >= 1 && c <= a.W && (unsigned)iy < (unsigned)a.H;
            dma16_buf(xr, ok ? (unsigned)((((b * a.H + iy) * a.W) + c - 1) * (a.x_ld * 2) + s * 16) : Y5_BUF_OOB, (unsigned)(d * 1024));
        }
    }
    const i32x4 wr = make_buf(a.w2, (C - 1) * a.w2_ld * 2 + ROW);
#pragma unroll
    for (int j = 0; j < W2_PIECES / NW; ++j) {
        const int d = wave + NW * j, row = d * 4 + (lane >> 4), sl = (lane & 15) ^ (row & 15);
        dma16_buf(wr, (unsigned)(((row & ~31) + perm_row(row & 31)) * (a.w2_ld * 2) + sl * 16), (unsigned)(W20 + d * 1024));
    }
    C128_STAMP(1)
    float pv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {  // [b1 | s1 | h1 | b2 | s2 | h2][C]
        const int i = t + j * NTH, sel = i >> 7, ch = i & (C - 1);
        const float* src = sel == 0 ? a.b1 : sel == 1 ? a.s1 : sel == 2 ? (a.s1 ? a.h1 : nullptr) : sel == 3 ? a.b2 : sel == 4 ? a.s2 : (a.s2 ? a.h2 : nullptr);
        pv[j] = (i < 6 * C && src) ? src[ch] : (sel == 1 || sel == 4) ? 1.f : 0.f;
    }
    // The 3x3 weights are fetched as whole 128-byte lines (a load instruction covers eight rows' 64 channels of one tap:
    // fetched the way the matrix instruction wants them, 32 rows x 32 bytes per instruction, the loads cost the CU four times
    // as many L2 requests and 12 k cycles to issue) and brought into fragment order through LDS below.
    f16x8 wf[36];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            wf[tap * 4 + j] = *reinterpret_cast<const f16x8*>(a.w1 + (size_t)(cb * 32 + 8 * j + (lane >> 3)) * a.w1_ld + tap * C + kh * 64 + (lane & 7) * 8);
        asm volatile("" ::: "memory");  // keep the taps in this order: the permutation below consumes them in it
    }
    C128_STAMP(0)
    float* prm = reinterpret_cast<float*>(smem + PRM0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
        if (t + j * NTH < 6 * C) prm[t + j * NTH] = pv[j];
    float negzero;
    asm volatile("v_mov_b32 %0, 0x80000000" : "=v"(negzero));
    {   // fragment order: register tap * 4 + ks of lane (lr, lh) holds channels [64 kh + 16 ks + 8 lh, +8) of row perm(lr).  Four
        // registers (one tap: 32 rows x 128 bytes) at a time through 4 KiB of this wave's own in the hand-over area, which
        // nothing uses before the first barrier; chunk c of row r sits at slot c ^ (r & 7).  (The waits for the loads are the
        // compiler's: one per tap, each for that tap's four registers only.)
        unsigned char* stg = smem + RED0 + wave * 4096;
        const int rho = perm_row(lr);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 8 * j + (lane >> 3);
                *reinterpret_cast<f16x8*>(stg + ((r * 8 + ((lane & 7) ^ (r & 7))) << 4)) = wf[tap * 4 + j];
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wf[tap * 4 + ks] = *reinterpret_cast<const f16x8*>(stg + ((rho * 8 + ((2 * ks + lh) ^ (rho & 7))) << 4));
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the copies are older than the weights: long landed
    C128_STAMP(2)
    asm volatile("s_barrier" ::: "memory");  // strip, weights and parameters are in place
    C128_STAMP(3)

    for (int p0 = 0; p0 < npx; p0 += PASS) {
        if (npx - p0 > 32)
            pass<2, F1, F2>(a, smem, wf, p0, npx, pix0, lr, lh, lane, cb, kh, wave, keep_first, negzero C128_STAMP_PASS);
        else
            pass<1, F1, F2>(a, smem, wf, p0, npx, pix0, lr, lh, lane, cb, kh, wave, keep_first, negzero C128_STAMP_PASS);
    }
#ifdef Y5_STAMPS
    if (lane == 0 && blockIdx.x < 256) {
        unsigned long long* o = y5_c128_stamp_buf + (blockIdx.x * 8 + wave) * 12;
        for (int i = 0; i < 9; ++i) o[i] = st_acc[i];
        o[11] = 1;
    }
#endif
}

template <int F1, int F2>
int launch(const C128Args& a, int keep_first, unsigned grid, hipStream_t s) {
    static std::atomic<unsigned long long> attr_set{0};
    const unsigned long long bit = 1ull << y5_current_device();
    if (!(attr_set.load() & bit)) {
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_c128_kernel<F1, F2>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        attr_set.fetch_or(bit);
    }
    hipLaunchKernelGGL((conv_c128_kernel<F1, F2>), dim3(grid), dim3(NTH), (size_t)SMEM, s, a, keep_first);
    Y5_LAUNCH_CHECK("conv_c128_kernel");
    return Y5_OK;
}

}  // namespace

#ifdef Y5_STAMPS
extern "C" int y5_debug_stamps_c128(unsigned long long* dst, int n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(y5_c128_stamp_buf), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif

// f, g: descriptors validated by y5_launch_conv_chain.  Returns 0 when the pair is outside what this kernel is compiled for
// (then *rc is untouched), 1 after a launch attempt with its status in *rc.
int y5_try_conv_c128_chain(const y5_conv_desc& f, const y5_conv_desc& g, int keep_first, hipStream_t s, int* rc) {
    if (f.ksize != 3 || f.stride != 1 || f.cin != C || f.cout != C || g.cin != C || g.cout != C) return 0;
    if (f.w_in > PW - 2 || f.w_rows < C || g.w_rows < C || f.kpad < 9 * C || g.kpad < C) return 0;
    const long long px = (long long)f.batch * f.h * f.w_in;
    const long long xb = ((px - 1) * f.x_ld + C) * 2;
    if (xb >= 0x7fffffffLL) return 0;
    C128Args a{};
    a.x = static_cast<const f16*>(f.x);
    a.res = static_cast<const f16*>(f.res);
    a.y1 = static_cast<f16*>(f.y);
    a.y2 = static_cast<f16*>(g.y);
    a.w1 = static_cast<const f16*>(f.w);
    a.w2 = static_cast<const f16*>(g.w);
    a.b1 = f.bias; a.s1 = f.scale; a.h1 = f.shift; a.b2 = g.bias; a.s2 = g.scale; a.h2 = g.shift;
    a.w1_ld = f.kpad; a.w2_ld = g.kpad;
    a.H = f.h; a.W = f.w_in; a.x_ld = f.x_ld; a.res_ld = f.res_ld; a.y1_ld = f.y_ld; a.y2_ld = g.y_ld;
    a.x_bytes = (int)xb;
    a.strips_per_image = (f.h + R - 1) / R;
    const long long total = (long long)f.batch * a.strips_per_image;
    if (total <= 0 || total > 0x7fffffffLL) return 0;
    a.div_w = make_fastdiv((unsigned)f.w_in);
    a.div_spi = make_fastdiv((unsigned)a.strips_per_image);
    a.slope1 = f.act == Y5_ACT_LEAKY ? f.slope : 1.f;
    a.slope2 = g.act == Y5_ACT_LEAKY ? g.slope : 1.f;
    const int f1 = (f.bias ? 1 : 0) | (f.scale ? 2 : 0), f2 = (g.bias ? 1 : 0) | (g.scale ? 2 : 0);
    if (f1 == 1 && f2 == 1) *rc = launch<1, 1>(a, keep_first, (unsigned)total, s);
    else if (f1 == 1 && f2 == 2) *rc = launch<1, 2>(a, keep_first, (unsigned)total, s);
    else if (f1 == 2 && f2 == 2) *rc = launch<2, 2>(a, keep_first, (unsigned)total, s);
    else return 0;
    return 1;
}
