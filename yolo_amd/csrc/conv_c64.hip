// 3x3 stride-1 convolution 64 -> 64 (+ residual) followed by a 1x1 convolution 64 -> 64, as one PERSISTENT kernel: the
// Bottleneck.cv2 of a 64-channel BottleneckCSP chained to the next Bottleneck's cv1 or to the block's cv3
// (models/common.py:43, :61).  Same contract as the other y5_conv_chain_f16 variants.
//
// At 64 channels the whole 3x3 weight tensor (72 KiB) fits LDS beside the pixel tiles, so it is staged ONCE per
// workgroup and the kernel walks 8 x 16 pixel tiles: a loader wave copies the next tile's input (tile + one-pixel halo,
// [pixel][channel] rows with XOR-swizzled 16-byte slots) and residual tile global -> LDS by DMA while eight compute
// waves - one per (two tile rows, 32 output channels) - multiply the nine taps straight from the staged halo tile: no
// im2col re-fetch (every input pixel is read from memory about 1.4 times instead of 9), no per-tile weight traffic.  The
// first result passes through LDS to the 1x1 convolution, whose weights live in registers.  Barriers are LDS-only, the
// compute waves never wait for their own global stores, and the only vmcnt wait is the loader's.
#include <atomic>

#include "y5_common.h"

namespace {

#include "conv_common.h"
#include "epi_mix.h"

struct C64Args {
    const f16 *x, *res;
    f16 *y1, *y2;
    const f16 *w1, *w2;
    const float *b1, *s1, *h1, *b2, *s2, *h2;
    int w1_ld, w2_ld;
    int H, W, x_ld, res_ld, y1_ld, y2_ld, x_bytes, res_bytes;
    int tiles_y, ntiles;
    FastDiv div_tpi, div_ty;
    float slope1, slope2;
};

constexpr int C = 64;
constexpr int TH = 8, TW = 16, PW = TW + 2, PH = TH + 2, HP = PH * PW;  // tile, tile + halo (180 pixels)
constexpr int NWC = 8, NTHC = 64 * NWC, NTH = NTHC + 64;                // compute waves + the loader wave
constexpr int ROW = C * 2;                                              // bytes per staged pixel (8 slots; halo tile: slot s at s ^ ((column >> 1) & 7), 16-pixel-row tiles: s ^ ((p >> 1) & 7), the same thing there)
constexpr int X_PIECES = (HP * ROW + 1023) / 1024, X_BYTES = X_PIECES * 1024;
constexpr int T_BYTES = TH * TW * ROW, T_PIECES = T_BYTES / 1024;       // residual tile / first result
constexpr int W1ROW = 9 * C * 2 + 16;                                   // weight row padded to an odd number of slots
constexpr int W10 = 0, X0 = W10 + C * W1ROW, RES0 = X0 + 2 * X_BYTES, PRM0 = RES0 + 2 * T_BYTES;  // two residual tiles: see V below
constexpr int SMEM = PRM0 + 2 * 3 * C * 4;
static_assert(X0 % 1024 == 0 && RES0 % 1024 == 0, "DMA pieces land on 1 KiB boundaries");
static_assert(SMEM <= 160 * 1024, "fits the CU's LDS");

__device__ __forceinline__ int swz(int p, int s) { return p * ROW + ((s ^ ((p >> 1) & 7)) << 4); }
// halo tile (rows of PW = 18 pixels): the swizzle follows the COLUMN, so that the 16 lanes one ds_read_b128 pass serves (eight
// pixels of one tile row, eight of the next, shifted by the tap) land on 16 different 16-byte bank groups for every tap
__device__ __forceinline__ int swzx(int r, int c, int s) { return (r * PW + c) * ROW + ((s ^ ((c >> 1) & 7)) << 4); }

#ifdef Y5_STAMPS
// Diagnostic build only: per compute wave, cycles in {barrier A, taps, first epilogue, barrier B, 1x1 + store} summed over tiles
__device__ unsigned long long y5_c64_stamp_buf[256 * 8 * 8];
#define C64_STAMP(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[i] += now_ - st_last; st_last = now_; }
#else
#define C64_STAMP(i)
#endif

// F1 / F2: post-ops of the two convolutions (bit 0 bias, bit 1 scale + shift), see epi_mix.h
template <int F1, int F2>
__global__ __launch_bounds__(NTH) void conv_c64_kernel(const C64Args a, const int keep_first) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, lr = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nwg = gridDim.x, g = xcd_remap(blockIdx.x, nwg);
    const int first = (int)((long long)g * a.ntiles / nwg), last = (int)((long long)(g + 1) * a.ntiles / nwg);

    // the 3x3 weights (row i = output channel (i & 32) + perm(i & 31)) and the per-channel parameters, once
    if (wave < NWC) {
        // The compute waves stage the 3x3 weights and the per-channel parameters (the loader wave goes straight to the first
        // tile's copies).  All of a thread's loads are issued before the first is waited for: as a load -> store loop the
        // staging was eight dependent round trips to L2 / the Infinity Cache at the start of every launch, and the six
        // parameter vectors another four.
        constexpr int SLOTS = C * (9 * C / 8), ROUNDS = SLOTS / NTHC;
        static_assert(SLOTS % NTHC == 0 && 6 * C <= NTHC, "whole rounds of the compute waves");
        f16x8 stage[ROUNDS];
#pragma unroll
        for (int j = 0; j < ROUNDS; ++j) {
            const int i = t + j * NTHC, row = i / (9 * C / 8), sl = i - row * (9 * C / 8);
            stage[j] = *reinterpret_cast<const f16x8*>(a.w1 + (size_t)((row & 32) + perm_row(row & 31)) * a.w1_ld + sl * 8);
        }
        // [b1 | s1 | h1 | b2 | s2 | h2][C]: one value per thread
        const int sel = t / C, ch = t - sel * C;
        const float* src = sel == 0 ? a.b1 : sel == 1 ? a.s1 : sel == 2 ? (a.s1 ? a.h1 : nullptr) : sel == 3 ? a.b2 : sel == 4 ? a.s2 : (a.s2 ? a.h2 : nullptr);
        const float pv = (t < 6 * C && src) ? src[ch] : (sel == 1 || sel == 4) ? 1.f : 0.f;
#pragma unroll
        for (int j = 0; j < ROUNDS; ++j) {
            const int i = t + j * NTHC, row = i / (9 * C / 8), sl = i - row * (9 * C / 8);
            *reinterpret_cast<f16x8*>(smem + W10 + row * W1ROW + sl * 16) = stage[j];
        }
        if (t < 6 * C) reinterpret_cast<float*>(smem + PRM0)[t] = pv;
    }
    float* prm = reinterpret_cast<float*>(smem + PRM0);

    int tb, ty0, tx0;
    auto locate = [&](int tile) {
        tb = (int)fdiv((unsigned)tile, a.div_tpi);
        const int rem = tile - tb * (int)a.div_tpi.d;
        const int tx = (int)fdiv((unsigned)rem, a.div_ty);
        ty0 = (rem - tx * a.tiles_y) * TH;
        tx0 = tx * TW;
    };

    if (wave == NWC) {
        // ------------------------------------------------------------------ loader wave
        const i32x4 xr = make_buf(a.x, a.x_bytes);
        const i32x4 rr = make_buf(a.res ? a.res : a.x, a.res ? a.res_bytes : 0);
        // per-lane constants of the copies: byte offset from the tile's first (halo) pixel, and which halo row / column the
        // lane's 16 bytes belong to (bit 16: beyond the staged tile)
        int xrel[X_PIECES], xrc[X_PIECES], rrel[T_PIECES];
#pragma unroll
        for (int d = 0; d < X_PIECES; ++d) {
            const int slot = d * 64 + lane, hp = slot >> 3;
            const int r = hp / PW, c = hp - r * PW, s = (slot & 7) ^ ((c >> 1) & 7);
            xrel[d] = hp >= HP ? 0x7fffffff : (r * a.W + c) * (a.x_ld * 2) + s * 16;  // origin + this: below 2^32, at or above x_bytes
            xrc[d] = r | (c << 8) | (hp >= HP ? 1 << 16 : 0);
        }
#pragma unroll
        for (int d = 0; d < T_PIECES; ++d) {
            const int slot = d * 64 + lane, q = slot >> 3, s = (slot & 7) ^ ((q >> 1) & 7);
            const int r = q / TW, c = q - r * TW;
            rrel[d] = (r * a.W + c) * (a.res_ld * 2) + s * 16;
        }
        __builtin_amdgcn_s_setprio(3);  // the copies are issued as soon as the barrier releases them
        auto issue_x = [&](int tile, int buf) {
            locate(tile);
            const int base = ((tb * a.H + ty0 - 1) * a.W + tx0 - 1) * (a.x_ld * 2);
            const bool inner = ty0 > 0 && ty0 + TH < a.H && tx0 > 0 && tx0 + TW < a.W;  // the halo stays inside the image
            if (inner) {  // no clipping: tile origin + the lane's fixed offset (slots past the tile carry an out-of-range offset)
#pragma unroll
                for (int d = 0; d < X_PIECES; ++d) dma16_buf(xr, (unsigned)(base + xrel[d]), (unsigned)(X0 + buf * X_BYTES + d * 1024));
                return;
            }
#pragma unroll
            for (int d = 0; d < X_PIECES; ++d) {
                const int r = xrc[d] & 255, c = (xrc[d] >> 8) & 255;
                const bool ok = !(xrc[d] >> 16) && (unsigned)(ty0 - 1 + r) < (unsigned)a.H && (unsigned)(tx0 - 1 + c) < (unsigned)a.W;
                dma16_buf(xr, ok ? (unsigned)(base + xrel[d]) : Y5_BUF_OOB, (unsigned)(X0 + buf * X_BYTES + d * 1024));
            }
        };
        auto issue_res = [&](int tile, int buf) {
            if (!a.res) return;
            locate(tile);
            const int base = ((tb * a.H + ty0) * a.W + tx0) * (a.res_ld * 2);
#pragma unroll
            for (int d = 0; d < T_PIECES; ++d) dma16_buf(rr, (unsigned)(base + rrel[d]), (unsigned)(RES0 + buf * T_BYTES + d * 1024));
        };
        if (first < last) {
            issue_x(first, 0);
            issue_res(first, 0);
        }
        int cur = 0;
#ifdef Y5_STAMPS
        unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#endif
        for (int tile = first; tile < last; ++tile, cur ^= 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            C64_STAMP(5)  // waiting for the copies
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // A: input and residual tiles have landed
            C64_STAMP(7)
            if (tile + 1 < last) {  // both buffers were last read while tile - 1 was computed
                issue_x(tile + 1, cur ^ 1);
                issue_res(tile + 1, cur ^ 1);
            }
            C64_STAMP(6)  // issuing the copies
            asm volatile("s_barrier" ::: "memory");                                   // B
            C64_STAMP(7)
        }
#ifdef Y5_STAMPS
        if (lane == 0 && blockIdx.x < 256) {  // the loader's figures ride in the unused fields of compute wave 0's row
            unsigned long long* o = y5_c64_stamp_buf + (blockIdx.x * 8) * 8;
            o[5] = st_acc[5];
            o[6] = st_acc[6];
        }
#endif
        return;
    }

    // ---------------------------------------------------------------------- compute waves
    const int nt = wave >> 1, mt = wave & 1;  // tile rows 2 nt, 2 nt + 1; output channels [32 mt, +32)
    const int r = 2 * nt + (lr >> 4), c = lr & 15, q = nt * 32 + lr;
    f16x8 w2f[C / 16];
#pragma unroll
    for (int ks = 0; ks < C / 16; ++ks) w2f[ks] = *reinterpret_cast<const f16x8*>(a.w2 + (size_t)(mt * 32 + perm_row(lr)) * a.w2_ld + ks * 16 + lh * 8);
    int uoff[9];  // byte offset in the staged input of this lane's K-step-0 fragment per tap; K-step ks: offset ^ (ks << 5)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) uoff[tap] = swzx(r + tap / 3, c + tap % 3, lh);
    const unsigned char* w1p = smem + W10 + (mt * 32 + lr) * W1ROW + lh * 16;
    const int toff = swz(q, mt * 4 + 2 * lh);   // this lane's 16 channels [32 mt + 16 lh, +16) of pixel q: slots toff, toff ^ 16
    const int voff = swz(q, lh);                // first-result fragments of pixel q: K-step ks at voff ^ (ks << 5)
    float negzero;
    asm volatile("v_mov_b32 %0, 0x80000000" : "=v"(negzero));
    int cur = 0;
#ifdef Y5_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#endif
    for (int tile = first; tile < last; ++tile, cur ^= 1) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // A
        C64_STAMP(0)
        locate(tile);
        const unsigned char* X = smem + X0 + cur * X_BYTES;
        // The first result overwrites the residual tile in place: a lane stores exactly the 32 bytes it has just read.
        unsigned char* V = smem + RES0 + cur * T_BYTES;
        f32x16 acc = zero16();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < C / 16; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(w1p + (tap * (C / 16) + ks) * 32),
                                                             *reinterpret_cast<const f16x8*>(X + (uoff[tap] ^ (ks << 5))), acc, 0, 0, 0);
        C64_STAMP(1)
        f16x8 lo, hi;
        finish<F1>(acc, prm, C, mt * 32 + 16 * lh, a.slope1, negzero, lo, hi);
        if (a.res) {
            lo += *reinterpret_cast<const f16x8*>(V + toff);
            hi += *reinterpret_cast<const f16x8*>(V + (toff ^ 16));
        }
        *reinterpret_cast<f16x8*>(V + toff) = lo;
        *reinterpret_cast<f16x8*>(V + (toff ^ 16)) = hi;
        const size_t pix = (size_t)(tb * a.H + ty0 + r) * a.W + tx0 + c;
        if (keep_first) {
            f16* yp = a.y1 + pix * a.y1_ld + mt * 32 + 16 * lh;
            *reinterpret_cast<f16x8*>(yp) = lo;
            *reinterpret_cast<f16x8*>(yp + 8) = hi;
        }
        C64_STAMP(2)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // B: the first result is complete
        C64_STAMP(3)
        acc = zero16();
#pragma unroll
        for (int ks = 0; ks < C / 16; ++ks)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2f[ks], *reinterpret_cast<const f16x8*>(V + (voff ^ (ks << 5))), acc, 0, 0, 0);
        finish<F2>(acc, prm + 3 * C, C, mt * 32 + 16 * lh, a.slope2, negzero, lo, hi);
        f16* yp = a.y2 + pix * a.y2_ld + mt * 32 + 16 * lh;
        *reinterpret_cast<f16x8*>(yp) = lo;
        *reinterpret_cast<f16x8*>(yp + 8) = hi;
        C64_STAMP(4)
    }
#ifdef Y5_STAMPS
    if (lane == 0 && blockIdx.x < 256) {
        unsigned long long* o = y5_c64_stamp_buf + (blockIdx.x * 8 + wave) * 8;
        for (int i = 0; i < 5; ++i) o[i] = st_acc[i];
        o[7] = (unsigned long long)(last - first);
    }
#endif
}

int cu_count() {  // compute units of the current device (one persistent workgroup each)
    static std::atomic<int> cached[64];
    const int dev = y5_current_device();
    int n = cached[dev].load();
    if (n == 0) {
        int real = 0;
        if (hipGetDevice(&real) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, real) != hipSuccess || n <= 0) n = 256;
        cached[dev].store(n);
    }
    return n;
}

template <int F1, int F2>
int launch(const C64Args& a, int keep_first, hipStream_t s) {
    static std::atomic<unsigned long long> attr_set{0};
    const unsigned long long bit = 1ull << y5_current_device();
    if (!(attr_set.load() & bit)) {
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_c64_kernel<F1, F2>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        attr_set.fetch_or(bit);
    }
    const int cus = cu_count();
    hipLaunchKernelGGL((conv_c64_kernel<F1, F2>), dim3((unsigned)(a.ntiles < cus ? a.ntiles : cus)), dim3(NTH), (size_t)SMEM, s, a, keep_first);
    Y5_LAUNCH_CHECK("conv_c64_kernel");
    return Y5_OK;
}

}  // namespace

#ifdef Y5_STAMPS
extern "C" int y5_debug_stamps_c64(unsigned long long* dst, int n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(y5_c64_stamp_buf), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif

// f, g: descriptors validated by y5_launch_conv_chain.  Returns 0 when the pair is outside what this kernel is compiled for
// (then *rc is untouched), 1 after a launch attempt with its status in *rc.
int y5_try_conv_c64_chain(const y5_conv_desc& f, const y5_conv_desc& g, int keep_first, hipStream_t s, int* rc) {
    if (f.ksize != 3 || f.stride != 1 || f.cin != C || f.cout != C || g.cout != C) return 0;
    if (f.h % TH || f.w_in % TW) return 0;
    const int f1 = (f.bias ? 1 : 0) | (f.scale ? 2 : 0), f2 = (g.bias ? 1 : 0) | (g.scale ? 2 : 0);
    const long long px = (long long)f.batch * f.h * f.w_in;
    const long long xb = ((px - 1) * f.x_ld + C) * 2, rb = f.res ? ((px - 1) * f.res_ld + C) * 2 : 0;
    if (xb >= 0x7fffffffLL || rb >= 0x7fffffffLL) return 0;
    C64Args a{};
    a.x = static_cast<const f16*>(f.x);
    a.res = static_cast<const f16*>(f.res);
    a.y1 = static_cast<f16*>(f.y);
    a.y2 = static_cast<f16*>(g.y);
    a.w1 = static_cast<const f16*>(f.w);
    a.w2 = static_cast<const f16*>(g.w);
    a.b1 = f.bias; a.s1 = f.scale; a.h1 = f.shift; a.b2 = g.bias; a.s2 = g.scale; a.h2 = g.shift;
    a.w1_ld = f.kpad; a.w2_ld = g.kpad;
    a.H = f.h; a.W = f.w_in; a.x_ld = f.x_ld; a.res_ld = f.res_ld; a.y1_ld = f.y_ld; a.y2_ld = g.y_ld;
    a.x_bytes = (int)xb; a.res_bytes = (int)rb;
    a.tiles_y = f.h / TH;
    const int tiles_x = f.w_in / TW;
    const long long total = (long long)f.batch * a.tiles_y * tiles_x;
    if (total <= 0 || total > 0x7fffffffLL) return 0;
    a.ntiles = (int)total;
    a.div_tpi = make_fastdiv((unsigned)(a.tiles_y * tiles_x));
    a.div_ty = make_fastdiv((unsigned)a.tiles_y);
    a.slope1 = f.act == Y5_ACT_LEAKY ? f.slope : 1.f;
    a.slope2 = g.act == Y5_ACT_LEAKY ? g.slope : 1.f;
    if (f1 == 1 && f2 == 1) *rc = launch<1, 1>(a, keep_first, s);
    else if (f1 == 1 && f2 == 2) *rc = launch<1, 2>(a, keep_first, s);
    else if (f1 == 2 && f2 == 2) *rc = launch<2, 2>(a, keep_first, s);
    else return 0;
    return 1;
}
