// BottleneckCSP with one Bottleneck as ONE kernel (models/common.py:33-62): for an 8 x 32 pixel tile of the output
//     [t | y2] = post([W1; W2] . x)           1x1, on the tile + a one-pixel halo   (cv1 with its BatchNorm/activation, cv2 with
//                                                                                the second half of the concat BatchNorm)
//     u        = post(Wm1 . t)            1x1, tile + halo, zero outside the image (Bottleneck.cv1; the 3x3's zero padding)
//     v        = post(Wm2 * u) (+ t)      3x3, tile                              (Bottleneck.cv2 and the shortcut)
//     y1       = post(W3 . v)             1x1                                    (cv3 + first half of the concat BatchNorm)
//     out      = post(W4 . [y1 | y2])     1x1                                    (cv4)
// with every intermediate tensor in LDS only: the block reads its input once and writes its output once.  Each post() is
// the per-layer epilogue of y5_conv_f16 (bias, folded BatchNorm, LeakyReLU, each rounded to fp16), so the values are those
// of the layer-by-layer evaluation up to the fp32 summation order.
//
// Work split.  One persistent workgroup of 8 waves per CU walks a contiguous run of tiles.  The weights are MFMA A operands:
// the small 1x1 stages keep theirs in registers for the whole kernel, m.cv2 and cv4 read theirs from LDS, where they are
// staged once.  The pixel operands are read from LDS in the [pixel][channel] layout with XOR-swizzled 16-byte slots
// (conflict-free ds_read_b128 without padding; the padding would not fit).  The input tile
// of the NEXT step is copied global -> LDS by DMA while the current one is computed.  Per tile three barriers: input
// landed / t ready / u ready; the 3x3, cv3 and cv4 of a tile row are computed by the same wave from rows that only it
// writes, so no barrier separates them.
#include <atomic>

#include "y5_common.h"

namespace {

#include "conv_common.h"
#include "epi_mix.h"

struct CspStage {
    const f16* w;
    int w_ld;
    const float *bias, *scale, *shift;
};

struct CspArgs {
    const f16* x;
    f16* y;
    int H, W, x_ld, y_ld, x_bytes;
    int tiles_y, ntiles;
    FastDiv div_tpi, div_ty;
    CspStage s1, s2, sm1, sm2, s3, s4;
    float slope;
    int shortcut;
};

constexpr int TH = 8, TW = 32, HW = TW + 2, HH = TH + 2, HP = HH * HW;  // tile, tile + halo (340 pixels)
constexpr int NT1 = (HP + 31) / 32;                                     // 32-pixel groups of the halo tile (11)
constexpr int NWAVE = TH, NTHREADS = 64 * NWAVE;                        // one wave per tile row

// Which post-ops a stage has, as template bits: stage s (0 = cv1, 1 = cv2, 2 = m.cv1, 3 = m.cv2, 4 = cv3, 5 = cv4) has a
// bias when bit 2s is set and a scale + shift when bit 2s+1 is set.
constexpr int stage_flags(int s, int flags) { return (flags >> (2 * s)) & 3; }
constexpr int kFlagsFused = 1 | (2 << 2) | (1 << 4) | (1 << 6) | (2 << 8) | (1 << 10);  // Model.fuse()d: Conv = conv + bias
constexpr int kFlagsBn = 2 | (2 << 2) | (2 << 4) | (2 << 6) | (2 << 8) | (2 << 10);     // every conv followed by a BatchNorm

template <int C1, int CH, int C2>
struct CspCfg {
    static_assert(C1 == 64 && CH == 32 && C2 == 64, "compiled for the 64 -> (32) -> 64 block");  // two K-steps per 32 channels: see uoff
    // LDS rows are unpadded; 16-byte slot s of row p is stored at slot s ^ swz(p), which spreads the 8 rows a ds_read_b128
    // serves together over all banks: input pixels are 128 bytes (8 slots, swz = (p >> 1) & 7), pixels of t / u / v / y1 / y2
    // are 64 bytes (4 slots, swz = (p >> 2) & 3).
    static constexpr int XROW = C1 * 2, TROW = CH * 2;
    static constexpr int X_PIECES = (HP * XROW + 1023) / 1024;   // 1 KiB DMA pieces of one staged input tile (43)
    static constexpr int X_BYTES = NT1 * 32 * XROW;              // stage 1 reads whole groups of 32 pixels (45056)
    static_assert(X_PIECES * 1024 <= X_BYTES, "the copy stays inside its buffer");
    static constexpr int U_BYTES = NT1 * 32 * TROW, V_BYTES = TH * TW * TROW;
    static_assert(U_BYTES + V_BYTES <= X_BYTES, "u and v fit the dead input buffer");
    static constexpr int X0 = 0, T0 = 2 * X_BYTES, Y20 = T0 + U_BYTES, PRM0 = Y20 + V_BYTES;
    static constexpr int NPRM = 3 * (2 * CH + CH + CH + CH + C2);
    static constexpr int W40 = PRM0 + NPRM * 4;   // cv4: row i = channel (i & 32) + perm(i & 31), slot s at s ^ ((i >> 1) & 7)
    static constexpr int WM2ROW = 9 * CH * 2 + 16;  // m.cv2: row i = channel perm(i), padded to an odd number of slots
    static constexpr int WM20 = W40 + C2 * 2 * CH * 2;
    static constexpr int SMEM = WM20 + CH * WM2ROW;
    static_assert(2 * CH * 2 == 128 && W40 % 16 == 0 && WM20 % 16 == 0, "16-byte aligned weight blocks");
    static_assert(SMEM <= 160 * 1024, "fits the CU's LDS");
    // parameter blocks [bias | scale | shift] per stage, in floats from PRM0
    static constexpr int P1 = 0, P2 = P1 + 3 * CH, PM1 = P2 + 3 * CH, PM2 = PM1 + 3 * CH, P3 = PM2 + 3 * CH, P4 = P3 + 3 * CH;
};

__device__ __forceinline__ f16x8 load_wfrag(const CspStage& st, int row0, int k0, int lr, int lh) {
    return *reinterpret_cast<const f16x8*>(st.w + (size_t)(row0 + perm_row(lr)) * st.w_ld + k0 + 8 * lh);
}

// Per-channel parameters of a stage with n <= NTHREADS channels: fetched by one call, stored by another, so that the loads
// of all stages (and of the weights) are in flight together instead of one round trip after the other.
struct StageParams {
    float b, s, h;
};
__device__ __forceinline__ StageParams fetch_params(const CspStage& st, int n, int t) {
    StageParams v = {0.f, 1.f, 0.f};
    if (t < n) {
        if (st.bias) v.b = st.bias[t];
        if (st.scale) {
            v.s = st.scale[t];
            v.h = st.shift[t];
        }
    }
    return v;
}
__device__ __forceinline__ void store_params(float* prm, const StageParams& v, int n, int t) {
    if (t < n) {
        prm[t] = v.b;
        prm[n + t] = v.s;
        prm[2 * n + t] = v.h;
    }
}

// byte offset of logical 16-byte slot s of row p in an array of 64-byte rows
__device__ __forceinline__ int trow_off(int p, int s) { return p * 64 + ((s ^ ((p >> 2) & 3)) << 4); }

#ifdef Y5_STAMPS
// Diagnostic build only: per wave, cycles spent in {top wait, copy issue, stage 1, barrier, stage 2, barrier, rows} summed over tiles
__device__ unsigned long long y5_csp_stamp_buf[256 * NWAVE * 8];
#define CSP_STAMP(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[i] += now_ - st_last; st_last = now_; }
#else
#define CSP_STAMP(i)
#endif

template <int C1, int CH, int C2, int FLAGS>
__global__ __launch_bounds__(NTHREADS) void csp1_kernel(const CspArgs a) {
    typedef CspCfg<C1, CH, C2> Cfg;
    constexpr int K1 = C1 / 16, KH = CH / 16;
    constexpr int X_BYTES = Cfg::X_BYTES;
    constexpr int DMA_PER_WAVE = (Cfg::X_PIECES + NWAVE - 1) / NWAVE;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, lr = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);

    float* prm = reinterpret_cast<float*>(smem + Cfg::PRM0);
    float negzero;
    asm volatile("v_mov_b32 %0, 0x80000000" : "=v"(negzero));  // opaque to the optimiser: x*s + (-0) keeps the sign of a zero product

    // ---- the wave's share of the input copy: piece d = wave + 8j is 1 KiB of the staged tile, 16-byte slot d*64 + lane
    unsigned rc[DMA_PER_WAVE];  // halo row | halo column << 8 | nothing to copy << 20
    int rel[DMA_PER_WAVE];      // byte offset of the lane's 16 bytes from the halo tile's first pixel
#pragma unroll
    for (int j = 0; j < DMA_PER_WAVE; ++j) {
        const unsigned slot = (unsigned)((wave + NWAVE * j) * 64 + lane);
        const unsigned px = slot >> 3, sl = (slot & 7u) ^ ((px >> 1) & 7u);
        const unsigned r = px / HW, c = px - r * HW;
        rc[j] = r | (c << 8) | (px >= (unsigned)HP ? 1u << 20 : 0u);
        rel[j] = (int)((r * (unsigned)a.W + c) * (unsigned)(a.x_ld * 2) + sl * 16u);
    }
    const i32x4 xr = make_buf(a.x, a.x_bytes);

    const int nwg = gridDim.x, g = xcd_remap(blockIdx.x, nwg);
    const int first = (int)((long long)g * a.ntiles / nwg), last = (int)((long long)(g + 1) * a.ntiles / nwg);

    int tb, ty0, tx0;  // image and top-left pixel of a tile
    auto locate = [&](int tile) {
        tb = (int)fdiv((unsigned)tile, a.div_tpi);
        const int rem = tile - tb * (int)a.div_tpi.d;
        const int tx = (int)fdiv((unsigned)rem, a.div_ty);
        ty0 = (rem - tx * a.tiles_y) * TH;
        tx0 = tx * TW;
    };
    auto issue = [&](int tile, int buf) {
        locate(tile);
        const int base = ((tb * a.H + ty0 - 1) * a.W + tx0 - 1) * (a.x_ld * 2);
        const bool inner = ty0 > 0 && ty0 + TH < a.H && tx0 > 0 && tx0 + TW < a.W;  // the halo stays inside the image
#pragma unroll
        for (int j = 0; j < DMA_PER_WAVE; ++j) {
            if (wave + NWAVE * j >= Cfg::X_PIECES) break;  // wave-uniform
            bool ok = !(rc[j] >> 20);
            if (!inner) {
                const int r = (int)(rc[j] & 255u), c = (int)((rc[j] >> 8) & 255u);
                ok = ok && (unsigned)(ty0 - 1 + r) < (unsigned)a.H && (unsigned)(tx0 - 1 + c) < (unsigned)a.W;
            }
            dma16_buf(xr, ok ? (unsigned)(base + rel[j]) : Y5_BUF_OOB, (unsigned)(buf * X_BYTES + (wave + NWAVE * j) * 1024));
        }
    };
    // The first tile's copies go out before anything else is fetched, and everything the workgroup stages once is
    // requested before any of it is waited for (as load -> store loops the parameters and the LDS-resident weights were
    // ten dependent round trips to L2 / the Infinity Cache in front of the first copy).
    if (first < last) issue(first, 0);

    // ---- weights.  In registers for the whole kernel: this wave's half of [cv1 | cv2], m.cv1, cv3.  In LDS: m.cv2, cv4.
    const int mt1 = wave & 1;
    f16x8 w12[K1], wm1[KH], w3[KH];
#pragma unroll
    for (int ks = 0; ks < K1; ++ks) w12[ks] = load_wfrag(mt1 ? a.s2 : a.s1, 0, ks * 16, lr, lh);
#pragma unroll
    for (int ks = 0; ks < KH; ++ks) wm1[ks] = load_wfrag(a.sm1, 0, ks * 16, lr, lh);
#pragma unroll
    for (int ks = 0; ks < KH; ++ks) w3[ks] = load_wfrag(a.s3, 0, ks * 16, lr, lh);
    {
        static_assert(CH <= NTHREADS && C2 <= NTHREADS && C2 * 8 <= NTHREADS, "one parameter and one cv4 slot per thread");
        const StageParams q1 = fetch_params(a.s1, CH, t), q2 = fetch_params(a.s2, CH, t), qm1 = fetch_params(a.sm1, CH, t),
                          qm2 = fetch_params(a.sm2, CH, t), q3 = fetch_params(a.s3, CH, t), q4 = fetch_params(a.s4, C2, t);
        constexpr int M2_SLOTS = CH * (9 * CH / 8), M2_ROUNDS = (M2_SLOTS + NTHREADS - 1) / NTHREADS;
        f16x8 s4v = {}, m2v[M2_ROUNDS];
        if (t < C2 * 8) {
            const int row = t >> 3, sl = t & 7;
            s4v = *reinterpret_cast<const f16x8*>(a.s4.w + (size_t)((row & 32) + perm_row(row & 31)) * a.s4.w_ld + sl * 8);
        }
#pragma unroll
        for (int j = 0; j < M2_ROUNDS; ++j) {
            const int i = t + j * NTHREADS, row = i / (9 * CH / 8), sl = i - row * (9 * CH / 8);
            if (i < M2_SLOTS) m2v[j] = *reinterpret_cast<const f16x8*>(a.sm2.w + (size_t)perm_row(row) * a.sm2.w_ld + sl * 8);
        }
        store_params(prm + Cfg::P1, q1, CH, t);
        store_params(prm + Cfg::P2, q2, CH, t);
        store_params(prm + Cfg::PM1, qm1, CH, t);
        store_params(prm + Cfg::PM2, qm2, CH, t);
        store_params(prm + Cfg::P3, q3, CH, t);
        store_params(prm + Cfg::P4, q4, C2, t);
        if (t < C2 * 8) {
            const int row = t >> 3, sl = t & 7;
            *reinterpret_cast<f16x8*>(smem + Cfg::W40 + row * 128 + ((sl ^ ((row >> 1) & 7)) << 4)) = s4v;
        }
#pragma unroll
        for (int j = 0; j < M2_ROUNDS; ++j) {
            const int i = t + j * NTHREADS, row = i / (9 * CH / 8), sl = i - row * (9 * CH / 8);
            if (i < M2_SLOTS) *reinterpret_cast<f16x8*>(smem + Cfg::WM20 + row * Cfg::WM2ROW + sl * 16) = m2v[j];
        }
    }
    const unsigned char* w4p = smem + Cfg::W40 + lr * 128;
    const unsigned char* wm2p = smem + Cfg::WM20 + lr * Cfg::WM2ROW + lh * 16;
    int uoff[9];  // byte offset in u of this lane's K-step-0 fragment per tap (K-step 1 is the slot two further: offset ^ 32)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) uoff[tap] = trow_off((wave + tap / 3) * HW + lr + tap % 3, lh);

    int cur = 0;
#ifdef Y5_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#endif
    for (int tile = first; tile < last; ++tile, cur ^= 1) {
        // This wave's pieces of the tile are older than the four stores of the previous step: wait for them, then meet the
        // others (everyone is done with the previous step, whose u / v live in the buffer the next copy overwrites).
        if (tile == first)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        CSP_STAMP(0)
#ifdef Y5_STAMPS
        if (!(a.shortcut & 4))  // diagnostic: no copies behind the first
#endif
        if (tile + 1 < last) issue(tile + 1, cur ^ 1);
        locate(tile);
        CSP_STAMP(1)
        unsigned char* X = smem + cur * X_BYTES;  // the input tile; once t is computed, u (rows of the halo tile) and v
        unsigned char* T = smem + Cfg::T0;
        unsigned char* Y2 = smem + Cfg::Y20;
        unsigned char* U = X;
        unsigned char* V = X + Cfg::U_BYTES;

        // ---- [t | y2] on the halo tile: 2 * NT1 units of 32 pixels x 32 channels
        Prm16<stage_flags(0, FLAGS)> q1;
        Prm16<stage_flags(1, FLAGS)> q2;
        if (mt1 == 0) q1.load(prm + Cfg::P1, CH, 16 * lh); else q2.load(prm + Cfg::P2, CH, 16 * lh);
#pragma unroll
        for (int i = 0; i < (2 * NT1 + NWAVE - 1) / NWAVE; ++i) {
            const int u = wave + NWAVE * i;
            if (u >= 2 * NT1) break;
            const int px = (u >> 1) * 32 + lr;
            const unsigned char* bp = X + px * Cfg::XROW;
            f32x16 acc = zero16();
#pragma unroll
            for (int ks = 0; ks < K1; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w12[ks], *reinterpret_cast<const f16x8*>(bp + (((2 * ks + lh) ^ ((px >> 1) & 7)) << 4)), acc, 0, 0, 0);
            f16x8 lo, hi;
            if (mt1 == 0) {
                finish(acc, q1, a.slope, negzero, lo, hi);
                *reinterpret_cast<f16x8*>(T + trow_off(px, 2 * lh)) = lo;
                *reinterpret_cast<f16x8*>(T + trow_off(px, 2 * lh + 1)) = hi;
            } else {
                finish(acc, q2, a.slope, negzero, lo, hi);
                const int r = px / HW, c = px - r * HW;
                if (r >= 1 && r <= TH && c >= 1 && c <= TW) {
                    const int q = (r - 1) * TW + c - 1;
                    *reinterpret_cast<f16x8*>(Y2 + trow_off(q, 2 * lh)) = lo;
                    *reinterpret_cast<f16x8*>(Y2 + trow_off(q, 2 * lh + 1)) = hi;
                }
            }
        }
        CSP_STAMP(2)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        CSP_STAMP(3)

        // ---- u on the halo tile (zero where the halo leaves the image: the 3x3 convolution's padding)
#pragma unroll
        for (int i = 0; i < (NT1 + NWAVE - 1) / NWAVE; ++i) {
            const int nt = (NWAVE - 1 - wave) + NWAVE * i;  // the waves with one stage-1 unit less take two here
            if (nt >= NT1) break;
            const int px = nt * 32 + lr;
            f32x16 acc = zero16();
#pragma unroll
            for (int ks = 0; ks < KH; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wm1[ks], *reinterpret_cast<const f16x8*>(T + trow_off(px, 2 * ks + lh)), acc, 0, 0, 0);
            f16x8 lo, hi;
            finish<stage_flags(2, FLAGS)>(acc, prm + Cfg::PM1, CH, 16 * lh, a.slope, negzero, lo, hi);
            const int r = px / HW, c = px - r * HW;
            if (!((unsigned)(ty0 - 1 + r) < (unsigned)a.H && (unsigned)(tx0 - 1 + c) < (unsigned)a.W)) {
#pragma unroll
                for (int e = 0; e < 8; ++e) lo[e] = hi[e] = (f16)0.f;
            }
            *reinterpret_cast<f16x8*>(U + trow_off(px, 2 * lh)) = lo;
            *reinterpret_cast<f16x8*>(U + trow_off(px, 2 * lh + 1)) = hi;
        }
        CSP_STAMP(4)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        CSP_STAMP(5)

        // ---- tile row `wave`: v = post(Wm2 * u) + t, y1 = post(W3 . v), out = post(W4 . [y1 | y2])
        {
            f32x16 acc = zero16();
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
                for (int ks = 0; ks < KH; ++ks)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(wm2p + (tap * KH + ks) * 32),
                                                                 *reinterpret_cast<const f16x8*>(U + (uoff[tap] ^ (ks << 5))), acc, 0, 0, 0);
            }
            f16x8 lo, hi;
            finish<stage_flags(3, FLAGS)>(acc, prm + Cfg::PM2, CH, 16 * lh, a.slope, negzero, lo, hi);
            if (a.shortcut & 1) {
                const int p = (wave + 1) * HW + lr + 1;
                lo += *reinterpret_cast<const f16x8*>(T + trow_off(p, 2 * lh));
                hi += *reinterpret_cast<const f16x8*>(T + trow_off(p, 2 * lh + 1));
            }
            const int q = wave * TW + lr;
            *reinterpret_cast<f16x8*>(V + trow_off(q, 2 * lh)) = lo;
            *reinterpret_cast<f16x8*>(V + trow_off(q, 2 * lh + 1)) = hi;

            acc = zero16();
#pragma unroll
            for (int ks = 0; ks < KH; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3[ks], *reinterpret_cast<const f16x8*>(V + trow_off(q, 2 * ks + lh)), acc, 0, 0, 0);
            finish<stage_flags(4, FLAGS)>(acc, prm + Cfg::P3, CH, 16 * lh, a.slope, negzero, lo, hi);
            *reinterpret_cast<f16x8*>(V + trow_off(q, 2 * lh)) = lo;  // y1 over v: the same wave read these rows just above
            *reinterpret_cast<f16x8*>(V + trow_off(q, 2 * lh + 1)) = hi;

            f32x16 o0 = zero16(), o1 = zero16();
#pragma unroll
            for (int ks = 0; ks < 2 * KH; ++ks) {
                const f16x8 xf = *reinterpret_cast<const f16x8*>((ks < KH ? V : Y2) + trow_off(q, 2 * (ks % KH) + lh));
                const int so = ((2 * ks + lh) ^ ((lr >> 1) & 7)) << 4;
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(w4p + so), xf, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(w4p + 32 * 128 + so), xf, o1, 0, 0, 0);
            }
            f16* yp = a.y + ((size_t)(tb * a.H + ty0 + wave) * a.W + tx0 + lr) * a.y_ld + 16 * lh;
#ifdef Y5_STAMPS
            if (a.shortcut & 2) yp = a.y + (size_t)(blockIdx.x * NWAVE + wave) * 4096 + lane * 64;  // diagnostic: every store of a wave to one place
#endif
            finish<stage_flags(5, FLAGS)>(o0, prm + Cfg::P4, C2, 16 * lh, a.slope, negzero, lo, hi);
            *reinterpret_cast<f16x8*>(yp) = lo;
            *reinterpret_cast<f16x8*>(yp + 8) = hi;
            finish<stage_flags(5, FLAGS)>(o1, prm + Cfg::P4, C2, 32 + 16 * lh, a.slope, negzero, lo, hi);
            *reinterpret_cast<f16x8*>(yp + 32) = lo;
            *reinterpret_cast<f16x8*>(yp + 40) = hi;
        }
        CSP_STAMP(6)
    }
#ifdef Y5_STAMPS
    if (lane == 0 && blockIdx.x < 256) {
        unsigned long long* o = y5_csp_stamp_buf + (blockIdx.x * NWAVE + wave) * 8;
        for (int i = 0; i < 7; ++i) o[i] = st_acc[i];
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

CspStage make_stage(const y5_csp_stage& s) {
    return CspStage{static_cast<const f16*>(s.w), s.w_ld, s.bias, s.scale, s.shift};
}

}  // namespace

#ifdef Y5_STAMPS
extern "C" int y5_debug_stamps_csp(unsigned long long* dst, int n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(y5_csp_stamp_buf), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif

int y5_launch_csp(const y5_csp_desc& d, hipStream_t s) {
    Y5_REQUIRE(d.x && d.y && d.batch > 0 && d.h > 0 && d.w > 0, "csp: bad tensor arguments");
    Y5_REQUIRE(d.c1 == 64 && d.ch == 32 && d.c2 == 64, "csp: compiled for 64 -> (32) -> 64 channels, got %d -> (%d) -> %d", d.c1, d.ch, d.c2);
    Y5_REQUIRE(d.h % TH == 0 && d.w % TW == 0, "csp: height must be a multiple of %d and width of %d (got %d x %d)", TH, TW, d.h, d.w);
    Y5_REQUIRE(d.x_ld >= d.c1 && d.y_ld >= d.c2 && d.x_ld % 8 == 0 && d.y_ld % 8 == 0 && y5_aligned16(d.x) && y5_aligned16(d.y),
               "csp: tensors must be 16-byte aligned with pitches that are multiples of 8 channels");
    Y5_REQUIRE(d.slope > 0.f && d.slope <= 1.f, "csp: LeakyReLU slope %g outside (0, 1]", (double)d.slope);
    const y5_csp_stage* st[6] = {&d.cv1, &d.cv2, &d.m_cv1, &d.m_cv2, &d.cv3, &d.cv4};
    const int kmin[6] = {d.c1, d.c1, d.ch, 9 * d.ch, d.ch, 2 * d.ch};
    for (int i = 0; i < 6; ++i) {
        Y5_REQUIRE(st[i]->w && y5_aligned16(st[i]->w) && st[i]->w_ld >= kmin[i] && st[i]->w_ld % 8 == 0, "csp: stage %d weights missing, unaligned or too narrow", i);
        Y5_REQUIRE(!st[i]->scale || st[i]->shift, "csp: stage %d has a scale but no shift", i);
    }
    const long long xb = (((long long)d.batch * d.h * d.w - 1) * d.x_ld + d.c1) * 2;
    Y5_REQUIRE(xb < 0x7fffffffLL, "csp: input spans 2 GiB or more");
    CspArgs a{};
    a.x = static_cast<const f16*>(d.x);
    a.y = static_cast<f16*>(d.y);
    a.H = d.h; a.W = d.w; a.x_ld = d.x_ld; a.y_ld = d.y_ld; a.x_bytes = (int)xb;
    a.tiles_y = d.h / TH;
    const int tiles_x = d.w / TW;
    const long long total = (long long)d.batch * a.tiles_y * tiles_x;
    Y5_REQUIRE(total < 0x7fffffffLL, "csp: too many tiles");
    a.ntiles = (int)total;
    a.div_tpi = make_fastdiv((unsigned)(a.tiles_y * tiles_x));
    a.div_ty = make_fastdiv((unsigned)a.tiles_y);
    a.s1 = make_stage(d.cv1); a.s2 = make_stage(d.cv2); a.sm1 = make_stage(d.m_cv1); a.sm2 = make_stage(d.m_cv2); a.s3 = make_stage(d.cv3); a.s4 = make_stage(d.cv4);
    a.slope = d.slope;
    a.shortcut = d.shortcut;
    typedef CspCfg<64, 32, 64> Cfg;
    int flags = 0;
    for (int i = 0; i < 6; ++i) flags |= ((st[i]->bias ? 1 : 0) | (st[i]->scale ? 2 : 0)) << (2 * i);
    void (*kernel)(const CspArgs) = flags == kFlagsFused ? csp1_kernel<64, 32, 64, kFlagsFused> : flags == kFlagsBn ? csp1_kernel<64, 32, 64, kFlagsBn> : nullptr;
    Y5_REQUIRE(kernel, "csp: compiled for bias-only Convs (fused model) or scale+shift-only Convs (BatchNorm), got post-op pattern 0x%x", flags);
    static std::atomic<unsigned long long> attr_set[2];
    const unsigned long long bit = 1ull << y5_current_device();
    std::atomic<unsigned long long>& done = attr_set[flags == kFlagsFused ? 0 : 1];
    if (!(done.load() & bit)) {
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
        done.fetch_or(bit);
    }
    const int cus = cu_count();
    const unsigned grid = (unsigned)(total < cus ? total : cus);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(NTHREADS), (size_t)Cfg::SMEM, s, a);
    Y5_LAUNCH_CHECK("csp1_kernel");
    return Y5_OK;
}
