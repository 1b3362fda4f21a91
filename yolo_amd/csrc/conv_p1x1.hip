// 1x1 stride-1 convolution with 128 or 256 input channels and 128 or 256 output channels as a PERSISTENT kernel: the
// BottleneckCSP cv1 | cv2 and cv4 layers and the Bottleneck cv1 layers of the 40x40 stages (Conv.forward,
// models/common.py:26; BottleneckCSP.forward :59-62).  These layers move 26 - 52 MB for 1.7 - 6.7 GFLOP at batch 32: they
// are bound by memory, and a tiled GEMM kernel spends as long filling and draining its ring per tile as multiplying.
//
// One workgroup per CU walks a contiguous run of pixel tiles.  Eight waves - (output-channel blocks of 32) x (pixel groups)
// - keep the weights of their 32 channels in registers for the whole kernel (64 registers at 256 input channels), so
// after the first tile the only traffic is pixels in, pixels out.  Every wave copies four 1 KiB pieces of a tile's
// [pixel][channel] rows global -> LDS by DMA, two tiles ahead (three buffers); 16-byte slot s of pixel p is stored at
// s ^ (p & 15), which makes the fragment reads conflict-free.  One LDS-only barrier per tile; the one vmcnt wait per tile
// - for the copies issued a whole tile earlier, with the copies issued at the top of this tile allowed to stay in flight -
// sits right before the tile's stores, which go out from registers.
#include <atomic>

#include "y5_common.h"

namespace {

#include "conv_common.h"
#include "epi_mix.h"

struct P1Args {
    const f16* x;
    f16* y;
    const f16* w;
    const float *bias, *scale, *shift;
    int w_ld, x_ld, y_ld, x_bytes;
    int npix, ntiles;
    int cout;   // output channels with parameters behind them (conv_p1x1_kernel: the real width inside the 32 NCW layout)
    float slope;
};

constexpr int NWC = 8, NTH = 64 * NWC, NBUF = 3;

template <int KS, int NCW, int NT>  // 16 KS input channels, 32 NCW output channels, NT 32-pixel groups per wave and tile
struct P1Cfg {
    static constexpr int CI = 16 * KS, CO = 32 * NCW, PXW = NWC / NCW;
    static constexpr int TP = 32 * NT * PXW;                 // pixels per tile
    static constexpr int SLOTS = 2 * KS, XROW = SLOTS * 16;  // a staged pixel
    static_assert(SLOTS >= 16 && (SLOTS & (SLOTS - 1)) == 0, "the swizzle needs rows of 16 slots or more");
    static constexpr int X_BYTES = TP * XROW, PIECES = X_BYTES / 1024, MYP = PIECES / NWC;
    static_assert(MYP * NWC * 1024 == X_BYTES, "every wave copies the same number of whole pieces");
    static constexpr int PRM0 = NBUF * X_BYTES, SMEM = PRM0 + 3 * CO * 4;
    static_assert(SMEM <= 160 * 1024, "fits the CU's LDS");
};

// KSR <= KS and NCR <= NCW: the layer's REAL width inside the layout (160 input channels in rows laid out for 256: the
// swizzle needs a power-of-two number of slots per staged pixel; 160 output channels on the wave grid of 256).  Slots past
// the real width are never fetched (their lanes of a copy ask for an out-of-range address and get zeros), K-steps past it
// are not multiplied, waves whose channel block lies past it only copy.
template <int KS, int NCW, int NT, int F, int KSR = KS, int NCR = NCW>
__global__ __launch_bounds__(NTH) void conv_p1x1_kernel(const P1Args a) {
    typedef P1Cfg<KS, NCW, NT> Cfg;
    static_assert(KSR >= 1 && KSR <= KS && NCR >= 1 && NCR <= NCW, "real width inside the layout");
    constexpr int MYP = Cfg::MYP;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, lr = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // The workgroup's share: an equal, contiguous run of PIXELS (these layers are bound by memory: equal bytes per CU), walked
    // in tiles of TP pixels from its first pixel; the run's last tile may be partial.
    const int nwg = gridDim.x, g = xcd_remap(blockIdx.x, nwg);
    const int run0 = (int)((long long)g * a.npix / nwg), run1 = (int)((long long)(g + 1) * a.npix / nwg);
    const int first = 0, last = (run1 - run0 + Cfg::TP - 1) / Cfg::TP;

    float* prm = reinterpret_cast<float*>(smem + Cfg::PRM0);
    static_assert(Cfg::CO <= NTH, "one channel's parameters per thread");
    float pv[3] = {0.f, 1.f, 0.f};  // fetched here, stored once the weights have landed: one round trip for everything
    if (t < Cfg::CO && t < a.cout) {
        if (a.bias) pv[0] = a.bias[t];
        if (a.scale) {
            pv[1] = a.scale[t];
            pv[2] = a.shift[t];
        }
    }

    const int mt = wave % NCW, pg = wave / NCW;  // output channels [32 mt, +32); pixels [32 NT pg, +32 NT) of the tile
    const bool works = NCR == NCW || mt < NCR;   // wave-uniform: this wave's channel block exists

    // ---------------------------------------------------------------------- the copies: wave w issues pieces w, w + 8, ...
    const i32x4 xr = make_buf(a.x, a.x_bytes);
    int rel[MYP], cpx[MYP];  // per-lane constants: byte offset from the tile's first pixel, pixel within the tile
#pragma unroll
    for (int j = 0; j < MYP; ++j) {
        const int slot = (wave + NWC * j) * 64 + lane, px = slot / Cfg::SLOTS, sp = slot % Cfg::SLOTS;
        const int src = sp ^ (px & 15);           // the 16-byte piece of the pixel's channels that LDS slot sp holds
        cpx[j] = KSR == KS || src < 2 * KSR ? px : 0x7fffffff;   // pieces past the real width: never inside the run
        rel[j] = px * (a.x_ld * 2) + (src << 4);
    }
    auto issue = [&](int tile, int buf) {
        const int pix0 = run0 + tile * Cfg::TP, left = run1 - pix0;
        const int base = pix0 * (a.x_ld * 2);
#pragma unroll
        for (int j = 0; j < MYP; ++j)
            dma16_buf(xr, cpx[j] < left ? (unsigned)(base + rel[j]) : Y5_BUF_OOB, (unsigned)(buf * Cfg::X_BYTES + (wave + NWC * j) * 1024));
    };

    int uoff[NT];  // byte offset of this lane's K-step-0 fragment per pixel group; K-step ks: offset ^ (ks << 5)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int px = (pg * NT + nt) * 32 + lr;
        uoff[nt] = px * Cfg::XROW + ((lh ^ (px & 15)) << 4);
    }
    float negzero;
    asm volatile("v_mov_b32 %0, 0x80000000" : "=v"(negzero));

    // The first tile's pieces go out before the weights are fetched, so that both arrive in one round trip (parameters,
    // then weights, then copies cost the launch three).
    if (first < last) issue(first, 0);
    f16x8 wf[KSR];
#pragma unroll
    for (int ks = 0; ks < KSR; ++ks) wf[ks] = *reinterpret_cast<const f16x8*>(a.w + (size_t)((works ? mt : 0) * 32 + perm_row(lr)) * a.w_ld + ks * 16 + lh * 8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // from here on the only loads in flight are the copies counted below
#pragma unroll
    for (int ks = 0; ks < KSR; ++ks) asm volatile("" : "+v"(wf[ks]));  // in registers: nothing in the tile loop waits for them
    if (t < Cfg::CO) {
        prm[t] = pv[0];
        prm[Cfg::CO + t] = pv[1];
        prm[2 * Cfg::CO + t] = pv[2];
    }
    // two tiles ahead: when tile i's pieces are waited for, those of tile i + 1 (the newest MYP) may still be in flight
    if (first + 1 < last) issue(first + 1, 1);
    Prm16<F> q;
    bool have_prm = false;
    int buf = 0;
    for (int tile = first; tile < last; ++tile) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave's pieces of this tile have landed; tile - 1 is read
        if (tile + 2 < last) issue(tile + 2, buf == 0 ? NBUF - 1 : buf - 1);  // the buffer that held tile - 1
        if (!have_prm) {  // the parameters were staged by all threads before the first barrier
            q.load(prm, Cfg::CO, mt * 32 + 16 * lh);
            have_prm = true;
        }
        const unsigned char* X = smem + buf * Cfg::X_BYTES;
        f16x8 lo[NT], hi[NT];
        if (works) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x16 acc = zero16();
#pragma unroll
                for (int ks = 0; ks < KSR; ++ks)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks], *reinterpret_cast<const f16x8*>(X + (uoff[nt] ^ (ks << 5))), acc, 0, 0, 0);
                finish<F>(acc, q, a.slope, negzero, lo[nt], hi[nt]);
            }
        }
        // the next tile's pieces (issued a whole tile ago) must have landed before the next barrier; the pieces issued at the
        // top of this tile are younger and may stay in flight, this tile's stores are issued after the wait
        if (tile + 2 < last)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MYP) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int pix0 = run0 + tile * Cfg::TP;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int pix = pix0 + (pg * NT + nt) * 32 + lr;
            if (works && pix < run1) {
                f16* yp = a.y + (size_t)pix * a.y_ld + mt * 32 + 16 * lh;
                *reinterpret_cast<f16x8*>(yp) = lo[nt];
                *reinterpret_cast<f16x8*>(yp + 8) = hi[nt];
            }
        }
        buf = buf == NBUF - 1 ? 0 : buf + 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same kernel with a SECOND 1x1 convolution on the finished tile (BottleneckCSP's cv1 | cv2 followed by the first
// Bottleneck's cv1, which reads the cv1 half): the first result's leading
// 16 KS2 channels go from the registers into an LDS tile ([pixel][channel], slots swizzled like the input tiles), the
// eight waves regroup as (32 NCW2 output channels) x (32-pixel groups) and multiply that tile with the second layer's
// weights, which also stay in registers.  The first result is stored only if something else reads it (a.y != nullptr).
// Both stages apply bias, scale + shift and LeakyReLU from LDS-staged parameters (absent ones are 0 / 1 / 0, which
// leaves the roundings of the one-layer kernels unchanged).
struct P1ChainArgs {
    P1Args f;
    const f16* w2;
    const float *bias2, *scale2, *shift2;
    f16* y2;
    int w2_ld, y2_ld;
    float slope2;
};

template <int KS, int NCW, int KS2, int NCW2>
struct P1ChainCfg {
    static constexpr int CI = 16 * KS, CO = 32 * NCW, PXW = NWC / NCW;
    static constexpr int CI2 = 16 * KS2, CO2 = 32 * NCW2, PXW2 = NWC / NCW2;
    static_assert(CI2 <= CO && CI2 % 32 == 0, "the second layer reads leading whole channel blocks of the first result");
    static constexpr int TP = 32 * PXW2, NT = TP / (32 * PXW);  // one 32-pixel group per wave in the second stage
    static_assert(NT >= 1 && NT * 32 * PXW == TP, "the first stage covers the tile with whole pixel groups");
    static constexpr int SLOTS = 2 * KS, XROW = SLOTS * 16;
    static_assert(SLOTS >= 16 && (SLOTS & (SLOTS - 1)) == 0, "the swizzle needs rows of 16 slots or more");
    static constexpr int X_BYTES = TP * XROW, PIECES = X_BYTES / 1024, MYP = PIECES / NWC;
    static_assert(MYP * NWC * 1024 == X_BYTES, "every wave copies the same number of whole pieces");
    static constexpr int TROW = CI2 * 2, T_BYTES = TP * TROW;  // the first result's tile (two of them); 128-byte rows swizzle by pixel pairs
    static constexpr int T0 = NBUF * X_BYTES, PRM0 = T0 + 2 * T_BYTES, SMEM = PRM0 + 3 * (CO + CO2) * 4;
    static_assert(SMEM <= 160 * 1024, "fits the CU's LDS");
    __device__ static __forceinline__ int tswz(int px) { return TROW >= 256 ? (px & 15) : ((px >> 1) & 7); }
};

template <int KS, int NCW, int KS2, int NCW2>
__global__ __launch_bounds__(NTH) void conv_p1x1_chain_kernel(const P1ChainArgs c) {
    typedef P1ChainCfg<KS, NCW, KS2, NCW2> Cfg;
    constexpr int MYP = Cfg::MYP, NT = Cfg::NT;
    const P1Args& a = c.f;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, lr = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nwg = gridDim.x, g = xcd_remap(blockIdx.x, nwg);  // equal runs of pixels, as in conv_p1x1_kernel
    const int run0 = (int)((long long)g * a.npix / nwg), run1 = (int)((long long)(g + 1) * a.npix / nwg);
    const int first = 0, last = (run1 - run0 + Cfg::TP - 1) / Cfg::TP;

    float* prm = reinterpret_cast<float*>(smem + Cfg::PRM0);
    float* prm2 = prm + 3 * Cfg::CO;
    static_assert(Cfg::CO <= NTH && Cfg::CO2 <= NTH, "one channel's parameters per thread");
    float pv[3] = {0.f, 1.f, 0.f}, pv2[3] = {0.f, 1.f, 0.f};  // fetched here, stored once the weights have landed
    if (t < Cfg::CO) {
        if (a.bias) pv[0] = a.bias[t];
        if (a.scale) {
            pv[1] = a.scale[t];
            pv[2] = a.shift[t];
        }
    }
    if (t < Cfg::CO2) {
        if (c.bias2) pv2[0] = c.bias2[t];
        if (c.scale2) {
            pv2[1] = c.scale2[t];
            pv2[2] = c.shift2[t];
        }
    }

    const int mt = wave % NCW, pg = wave / NCW;      // first stage: output channels [32 mt, +32), pixels [32 NT pg, +32 NT)
    const int mt2 = wave % NCW2, pg2 = wave / NCW2;  // second stage: output channels [32 mt2, +32), pixels [32 pg2, +32)
    const i32x4 xr = make_buf(a.x, a.x_bytes);
    int rel[MYP], cpx[MYP];
#pragma unroll
    for (int j = 0; j < MYP; ++j) {
        const int slot = (wave + NWC * j) * 64 + lane, px = slot / Cfg::SLOTS, sp = slot % Cfg::SLOTS;
        cpx[j] = px;
        rel[j] = px * (a.x_ld * 2) + ((sp ^ (px & 15)) << 4);
    }
    auto issue = [&](int tile, int buf) {
        const int pix0 = run0 + tile * Cfg::TP, left = run1 - pix0;
        const int base = pix0 * (a.x_ld * 2);
#pragma unroll
        for (int j = 0; j < MYP; ++j)
            dma16_buf(xr, cpx[j] < left ? (unsigned)(base + rel[j]) : Y5_BUF_OOB, (unsigned)(buf * Cfg::X_BYTES + (wave + NWC * j) * 1024));
    };

    int uoff[NT], toff[NT];  // first stage: fragment offset in the input tile, and where the lane's 16 result channels go in T
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int px = (pg * NT + nt) * 32 + lr;
        uoff[nt] = px * Cfg::XROW + ((lh ^ (px & 15)) << 4);
        toff[nt] = Cfg::T0 + px * Cfg::TROW;
    }
    const int px2 = pg2 * 32 + lr;
    const int voff = Cfg::T0 + px2 * Cfg::TROW + ((lh ^ Cfg::tswz(px2)) << 4);  // second stage: K-step ks at voff ^ (ks << 5)
    float negzero;
    asm volatile("v_mov_b32 %0, 0x80000000" : "=v"(negzero));

    // the first tile's pieces go out before the weights are fetched: one round trip for parameters, weights and copies
    if (first < last) issue(first, 0);
    f16x8 wf[KS], wg[KS2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wf[ks] = *reinterpret_cast<const f16x8*>(a.w + (size_t)(mt * 32 + perm_row(lr)) * a.w_ld + ks * 16 + lh * 8);
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) wg[ks] = *reinterpret_cast<const f16x8*>(c.w2 + (size_t)(mt2 * 32 + perm_row(lr)) * c.w2_ld + ks * 16 + lh * 8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // from here on the only loads in flight are the copies counted below
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(wf[ks]));  // in registers: nothing in the tile loop waits for them
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) asm volatile("" : "+v"(wg[ks]));
    if (t < Cfg::CO) {
        prm[t] = pv[0];
        prm[Cfg::CO + t] = pv[1];
        prm[2 * Cfg::CO + t] = pv[2];
    }
    if (t < Cfg::CO2) {
        prm2[t] = pv2[0];
        prm2[Cfg::CO2 + t] = pv2[1];
        prm2[2 * Cfg::CO2 + t] = pv2[2];
    }
    if (first + 1 < last) issue(first + 1, 1);
    // The second stage runs one tile behind the first (two T buffers): one barrier per tile orders both, and the waves have
    // the second stage's products of tile i - 1 to issue beside the first stage's of tile i.
    auto second = [&](int tile) {
        f32x16 acc = zero16();
        const unsigned char* T = smem + ((tile & 1) ? Cfg::T_BYTES : 0);
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wg[ks], *reinterpret_cast<const f16x8*>(T + (voff ^ (ks << 5))), acc, 0, 0, 0);
        f16x8 l2, h2;
        finish<3>(acc, prm2, Cfg::CO2, mt2 * 32 + 16 * lh, c.slope2, negzero, l2, h2);
        const int pix = run0 + tile * Cfg::TP + px2;
        if (pix < run1) {
            f16* yp = c.y2 + (size_t)pix * c.y2_ld + mt2 * 32 + 16 * lh;
            *reinterpret_cast<f16x8*>(yp) = l2;
            *reinterpret_cast<f16x8*>(yp + 8) = h2;
        }
    };
    // The waves never wait for their own stores: the one vmcnt wait per tile sits right before the NEXT tile's barrier and
    // only asks for that tile's pieces (issued a whole iteration earlier) - everything issued after them may stay in flight:
    // the previous iteration's stores, the pieces issued at the top of this one, this iteration's stores.  (Only the last
    // tile of a run can be ragged and skip store instructions; no wait follows it.)
    const int su = a.y ? 2 * NT : 0;  // store instructions of a full tile's first result
    int buf = 0, after = 0;           // vector-memory instructions issued since the next tile's pieces
    for (int tile = first; tile < last; ++tile) {
        // this tile's pieces have landed and tile - 1's T is complete; tile - 1's input and tile - 2's T are read
        if (tile == first) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
            switch (after) {
#define Y5_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
                Y5_W(2) Y5_W(4) Y5_W(6) Y5_W(8) Y5_W(10) Y5_W(12) Y5_W(14) Y5_W(16)
#undef Y5_W
                default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
            }
        }
        // instructions issued after the pieces of tile + 1 (which went out at the top of the previous iteration, or before the
        // loop for the second tile): the previous iteration's stores, then this iteration's pieces and stores
        after = tile == first ? 0 : su + (tile - 1 > first ? 2 : 0);
        if (tile + 2 < last) {
            issue(tile + 2, buf == 0 ? NBUF - 1 : buf - 1);
            after += MYP;
        }
        after += su + (tile > first ? 2 : 0);
        const unsigned char* X = smem + buf * Cfg::X_BYTES;
        unsigned char* T = smem + ((tile & 1) ? Cfg::T_BYTES : 0);
        f16x8 lo[NT], hi[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 acc = zero16();
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks], *reinterpret_cast<const f16x8*>(X + (uoff[nt] ^ (ks << 5))), acc, 0, 0, 0);
            finish<3>(acc, prm, Cfg::CO, mt * 32 + 16 * lh, a.slope, negzero, lo[nt], hi[nt]);  // parameters from LDS: the registers hold weights
            if (mt * 32 < Cfg::CI2) {  // wave-uniform: this channel block feeds the second layer
                const int px = (pg * NT + nt) * 32 + lr, s0 = 4 * mt + 2 * lh, f = Cfg::tswz(px);
                *reinterpret_cast<f16x8*>(T + toff[nt] + ((s0 ^ f) << 4)) = lo[nt];
                *reinterpret_cast<f16x8*>(T + toff[nt] + (((s0 + 1) ^ f) << 4)) = hi[nt];
            }
        }
        if (a.y) {
            const int pix0 = run0 + tile * Cfg::TP;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int pix = pix0 + (pg * NT + nt) * 32 + lr;
                if (pix < run1) {
                    f16* yp = a.y + (size_t)pix * a.y_ld + mt * 32 + 16 * lh;
                    *reinterpret_cast<f16x8*>(yp) = lo[nt];
                    *reinterpret_cast<f16x8*>(yp + 8) = hi[nt];
                }
            }
        }
        if (tile > first) second(tile - 1);
        buf = buf == NBUF - 1 ? 0 : buf + 1;
    }
    if (first < last) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the last tile's T is complete
        second(last - 1);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same kernel for a 1x1 convolution of the concatenation [nearest-2x-upsampled a | b] (nn.Upsample -> Concat -> the
// cv1 | cv2 of the head's BottleneckCSP, models/yolov5s.yaml head rows 11-13 / 15-17) with neither tensor built: 128
// half of the channels from pixel (oy / 2, ox / 2) of the half-resolution a, half from pixel (oy, ox) of b.  A staged tile
// is two planes of [pixel][channel] rows (slot s of pixel p at s ^ (p & 15)) - a 1 KiB copy then reads one
// source through one descriptor - and the first half of the K-steps multiplies from the first plane, the second half from
// the second.  Compiled for 128 + 128 -> 128 channels (80 x 80 in yolov5s) and 256 + 256 -> 256 (40 x 40: 128 weight
// registers per wave, tiles of 32 pixels).
struct P1CatArgs {
    P1Args f;          // x = a (half resolution), x_ld / x_bytes its pitch and span
    const f16* x2;     // b
    int x2_ld, x2_bytes;
    int H, W;          // output (= b's) height and width
    FastDiv div_hw, div_w;
};

template <int KSH, int NCW, int NT>  // 16 KSH channels per source
struct P1CatCfg {
    static constexpr int KS = 2 * KSH, CO = 32 * NCW, PXW = NWC / NCW;
    static constexpr int TP = 32 * NT * PXW;
    static constexpr int SLOTS = 2 * KSH, XROW = SLOTS * 16, PLANE = TP * XROW, X_BYTES = 2 * PLANE, PIECES = X_BYTES / 1024, MYP = PIECES / NWC;
    static_assert(SLOTS >= 16 && (SLOTS & (SLOTS - 1)) == 0, "the swizzle needs rows of 16 slots or more");
    static_assert(MYP * NWC * 1024 == X_BYTES && (PIECES / 2) % NWC == 0, "every wave copies the same number of whole pieces of each plane");
    static constexpr int PRM0 = NBUF * X_BYTES, SMEM = PRM0 + 3 * CO * 4;
    static_assert(SMEM <= 160 * 1024, "fits the CU's LDS");
};

template <int KSH, int NCW, int NT, int F>
__global__ __launch_bounds__(NTH) void conv_p1x1_cat_kernel(const P1CatArgs c) {
    typedef P1CatCfg<KSH, NCW, NT> Cfg;
    constexpr int MYP = Cfg::MYP, KS = Cfg::KS, HALF = Cfg::PIECES / 2;
    const P1Args& a = c.f;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, lr = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nwg = gridDim.x, g = xcd_remap(blockIdx.x, nwg);  // equal runs of pixels, as in conv_p1x1_kernel
    const int run0 = (int)((long long)g * a.npix / nwg), run1 = (int)((long long)(g + 1) * a.npix / nwg);
    const int first = 0, last = (run1 - run0 + Cfg::TP - 1) / Cfg::TP;

    float* prm = reinterpret_cast<float*>(smem + Cfg::PRM0);
    static_assert(Cfg::CO <= NTH, "one channel's parameters per thread");
    float pv[3] = {0.f, 1.f, 0.f};
    if (t < Cfg::CO) {
        if (a.bias) pv[0] = a.bias[t];
        if (a.scale) {
            pv[1] = a.scale[t];
            pv[2] = a.shift[t];
        }
    }
    const int mt = wave % NCW, pg = wave / NCW;

    // the copies: wave w issues pieces w, w + 8, ...; pieces below HALF fill the plane of a, the others the plane of b
    const i32x4 ar = make_buf(a.x, a.x_bytes), br = make_buf(c.x2, c.x2_bytes);
    int cpx[MYP], csl[MYP];  // per-lane constants: pixel within the tile, 16 x (source chunk of the lane's slot)
#pragma unroll
    for (int j = 0; j < MYP; ++j) {
        const int d = wave + NWC * j, dd = d < HALF ? d : d - HALF;
        const int slot = dd * 64 + lane, px = slot / Cfg::SLOTS, sp = slot % Cfg::SLOTS;
        cpx[j] = px;
        csl[j] = (sp ^ (px & 15)) << 4;
    }
    const int Ha = c.H >> 1, Wa = c.W >> 1;
    auto issue = [&](int tile, int buf) {
        const int pix0 = run0 + tile * Cfg::TP, left = run1 - pix0;
#pragma unroll
        for (int j = 0; j < MYP; ++j) {
            const int d = wave + NWC * j;  // wave-uniform: a piece reads one source
            const unsigned P = (unsigned)(pix0 + cpx[j]);
            unsigned off;
            if (d < HALF) {
                const unsigned b = fdiv(P, c.div_hw), rem = P - b * (unsigned)(c.H * c.W);
                const unsigned oy = fdiv(rem, c.div_w), ox = rem - oy * (unsigned)c.W;
                off = ((b * (unsigned)Ha + (oy >> 1)) * (unsigned)Wa + (ox >> 1)) * (unsigned)(a.x_ld * 2) + (unsigned)csl[j];
            } else {
                off = P * (unsigned)(c.x2_ld * 2) + (unsigned)csl[j];
            }
            dma16_buf(d < HALF ? ar : br, cpx[j] < left ? off : Y5_BUF_OOB, (unsigned)(buf * Cfg::X_BYTES + d * 1024));
        }
    };

    int uoff[NT];  // byte offset within a plane of this lane's K-step-0 fragment per pixel group; K-step k of the plane: offset ^ (k << 5)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int px = (pg * NT + nt) * 32 + lr;
        uoff[nt] = px * Cfg::XROW + ((lh ^ (px & 15)) << 4);
    }
    float negzero;
    asm volatile("v_mov_b32 %0, 0x80000000" : "=v"(negzero));

    if (first < last) issue(first, 0);  // before the weights are fetched: one round trip for parameters, weights and copies
    f16x8 wf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wf[ks] = *reinterpret_cast<const f16x8*>(a.w + (size_t)(mt * 32 + perm_row(lr)) * a.w_ld + ks * 16 + lh * 8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // from here on the only loads in flight are the copies counted below
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(wf[ks]));
    if (t < Cfg::CO) {
        prm[t] = pv[0];
        prm[Cfg::CO + t] = pv[1];
        prm[2 * Cfg::CO + t] = pv[2];
    }
    if (first + 1 < last) issue(first + 1, 1);  // two tiles ahead, as in conv_p1x1_kernel
    Prm16<F> q;
    bool have_prm = false;
    int buf = 0;
    for (int tile = first; tile < last; ++tile) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave's pieces of this tile have landed; tile - 1 is read
        if (tile + 2 < last) issue(tile + 2, buf == 0 ? NBUF - 1 : buf - 1);  // the buffer that held tile - 1
        if (!have_prm) {
            q.load(prm, Cfg::CO, mt * 32 + 16 * lh);
            have_prm = true;
        }
        const unsigned char* X = smem + buf * Cfg::X_BYTES;
        f16x8 lo[NT], hi[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x16 acc = zero16();
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks], *reinterpret_cast<const f16x8*>(X + (ks / KSH) * Cfg::PLANE + (uoff[nt] ^ ((ks % KSH) << 5))), acc, 0, 0, 0);
            finish<F>(acc, q, a.slope, negzero, lo[nt], hi[nt]);
        }
        if (tile + 2 < last)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MYP) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int pix0 = run0 + tile * Cfg::TP;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int pix = pix0 + (pg * NT + nt) * 32 + lr;
            if (pix < run1) {
                f16* yp = a.y + (size_t)pix * a.y_ld + mt * 32 + 16 * lh;
                *reinterpret_cast<f16x8*>(yp) = lo[nt];
                *reinterpret_cast<f16x8*>(yp + 8) = hi[nt];
            }
        }
        buf = buf == NBUF - 1 ? 0 : buf + 1;
    }
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

template <int KS, int NCW, int NT, int F, int KSR, int NCR>
int launch_f(P1Args& a, hipStream_t s) {
    typedef P1Cfg<KS, NCW, NT> Cfg;
    static std::atomic<unsigned long long> attr_set{0};
    const unsigned long long bit = 1ull << y5_current_device();
    if (!(attr_set.load() & bit)) {
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_p1x1_kernel<KS, NCW, NT, F, KSR, NCR>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
        attr_set.fetch_or(bit);
    }
    a.ntiles = (a.npix + Cfg::TP - 1) / Cfg::TP;
    const int cus = cu_count();
    hipLaunchKernelGGL((conv_p1x1_kernel<KS, NCW, NT, F, KSR, NCR>), dim3((unsigned)(a.ntiles < cus ? a.ntiles : cus)), dim3(NTH), (size_t)Cfg::SMEM, s, a);
    Y5_LAUNCH_CHECK("conv_p1x1_kernel");
    return Y5_OK;
}

template <int KS, int NCW, int NT, int KSR = KS, int NCR = NCW>
int launch(P1Args& a, int f, hipStream_t s) {
    return f == 0 ? launch_f<KS, NCW, NT, 0, KSR, NCR>(a, s) : f == 1 ? launch_f<KS, NCW, NT, 1, KSR, NCR>(a, s)
         : f == 2 ? launch_f<KS, NCW, NT, 2, KSR, NCR>(a, s) : launch_f<KS, NCW, NT, 3, KSR, NCR>(a, s);
}

template <int KS, int NCW, int KS2, int NCW2>
int launch_chain(P1ChainArgs& c, hipStream_t s) {
    typedef P1ChainCfg<KS, NCW, KS2, NCW2> Cfg;
    static std::atomic<unsigned long long> attr_set{0};
    const unsigned long long bit = 1ull << y5_current_device();
    if (!(attr_set.load() & bit)) {
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_p1x1_chain_kernel<KS, NCW, KS2, NCW2>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
        attr_set.fetch_or(bit);
    }
    c.f.ntiles = (c.f.npix + Cfg::TP - 1) / Cfg::TP;
    const int cus = cu_count();
    hipLaunchKernelGGL((conv_p1x1_chain_kernel<KS, NCW, KS2, NCW2>), dim3((unsigned)(c.f.ntiles < cus ? c.f.ntiles : cus)), dim3(NTH), (size_t)Cfg::SMEM, s, c);
    Y5_LAUNCH_CHECK("conv_p1x1_chain_kernel");
    return Y5_OK;
}

template <int KSH, int NCW, int NT, int F>
int launch_cat_f(P1CatArgs& c, hipStream_t s) {
    typedef P1CatCfg<KSH, NCW, NT> Cfg;
    static std::atomic<unsigned long long> attr_set{0};
    const unsigned long long bit = 1ull << y5_current_device();
    if (!(attr_set.load() & bit)) {
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_p1x1_cat_kernel<KSH, NCW, NT, F>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
        attr_set.fetch_or(bit);
    }
    c.f.ntiles = (c.f.npix + Cfg::TP - 1) / Cfg::TP;
    const int cus = cu_count();
    hipLaunchKernelGGL((conv_p1x1_cat_kernel<KSH, NCW, NT, F>), dim3((unsigned)(c.f.ntiles < cus ? c.f.ntiles : cus)), dim3(NTH), (size_t)Cfg::SMEM, s, c);
    Y5_LAUNCH_CHECK("conv_p1x1_cat_kernel");
    return Y5_OK;
}

}  // namespace

int y5_conv_p1x1_count() { return 2; }

// which = 1 (two 32-pixel groups per wave and tile) or 2 (one: tiles half as large, twice as many).  Returns 0 when the
// convolution is outside what this kernel is compiled for (then *rc is untouched), 1 after a launch attempt with its
// status in *rc.  d was validated by y5_launch_conv.
int y5_try_conv_p1x1(const y5_conv_desc& d, hipStream_t s, int which, int* rc) {
    if (which < 1 || which > 2 || d.ksize != 1 || d.stride != 1 || d.pad != 0 || d.res) return 0;
    const bool pow2 = (d.cin == 128 || d.cin == 256) && (d.cout == 128 || d.cout == 256);
    // widths inside a power-of-two layout: yolov5m (96, 192) and yolov5x (160) 1x1 layers
    const bool inner = (d.cin == 160 && d.cout == 160) || (d.cin == 192 && d.cout == 192) || (d.cin == 96 && d.cout == 96) ||
                       (d.cin == 192 && d.cout == 96) || (d.cin == 320 && d.cout == 160) || (d.cin == 384 && d.cout == 192);
    if (!pow2 && !inner) return 0;
    if (d.w_rows < (d.cout + 31) / 32 * 32 || d.kpad < d.cin) return 0;
    const long long npix = (long long)d.batch * d.h * d.w_in;
    const long long xb = ((npix - 1) * d.x_ld + d.cin) * 2;
    if (npix <= 0 || npix >= 0x7fffffffLL || xb >= 0x7fffffffLL) return 0;
    P1Args a{};
    a.x = static_cast<const f16*>(d.x);
    a.y = static_cast<f16*>(d.y);
    a.w = static_cast<const f16*>(d.w);
    a.bias = d.bias; a.scale = d.scale; a.shift = d.shift;
    a.w_ld = d.kpad; a.x_ld = d.x_ld; a.y_ld = d.y_ld; a.x_bytes = (int)xb;
    a.npix = (int)npix;
    a.cout = d.cout;
    a.slope = d.act == Y5_ACT_LEAKY ? d.slope : 1.f;
    const int f = (d.bias ? 1 : 0) | (d.scale ? 2 : 0);
    if (inner) {
        if (d.cin == 160 && d.cout == 160) *rc = which == 1 ? launch<16, 8, 2, 10, 5>(a, f, s) : launch<16, 8, 1, 10, 5>(a, f, s);
        else if (d.cin == 192 && d.cout == 192) *rc = which == 1 ? launch<16, 8, 2, 12, 6>(a, f, s) : launch<16, 8, 1, 12, 6>(a, f, s);
        else if (d.cin == 96 && d.cout == 96) *rc = which == 1 ? launch<8, 4, 2, 6, 3>(a, f, s) : launch<8, 4, 1, 6, 3>(a, f, s);
        else if (d.cin == 192 && d.cout == 96) {
            if (which == 1) return 0;
            *rc = launch<16, 4, 1, 12, 3>(a, f, s);
        } else {  // 320 -> 160 and 384 -> 192: rows of 64 slots, 32-pixel tiles
            if (which == 1) return 0;
            *rc = d.cin == 320 ? launch<32, 8, 1, 20, 5>(a, f, s) : launch<32, 8, 1, 24, 6>(a, f, s);
        }
        return 1;
    }
    if (d.cin == 256 && d.cout == 256) *rc = which == 1 ? launch<16, 8, 2>(a, f, s) : launch<16, 8, 1>(a, f, s);
    else if (d.cin == 128 && d.cout == 128) *rc = which == 1 ? launch<8, 4, 2>(a, f, s) : launch<8, 4, 1>(a, f, s);
    else if (d.cin == 256 && d.cout == 128) {
        if (which == 1) return 0;  // three 64 KiB buffers do not fit
        *rc = launch<16, 4, 1>(a, f, s);
    } else {
        *rc = which == 1 ? launch<8, 8, 2>(a, f, s) : launch<8, 8, 1>(a, f, s);
    }
    return 1;
}

// f followed by g, a 1x1 stride-1 convolution of the first g.cin channels of f's result (f's result is stored only if
// keep_first).  Compiled for f: 256 -> 256 with g: 128 -> 128, and f: 128 -> 128 with g: 64 -> 64.  Returns 0
// when the pair is outside that (then *rc is untouched), 1 after a launch attempt with its status in *rc.  Both
// descriptors were validated by y5_launch_conv_chain.
int y5_try_conv_p1x1_chain(const y5_conv_desc& f, const y5_conv_desc& g, int keep_first, hipStream_t s, int* rc) {
    if (f.ksize != 1 || f.stride != 1 || f.pad != 0 || f.res || g.res) return 0;
    const int shape = f.cin == 256 && f.cout == 256 && g.cin == 128 && g.cout == 128 ? 1 : f.cin == 128 && f.cout == 128 && g.cin == 64 && g.cout == 64 ? 3 : 0;
    if (!shape) return 0;
    if (f.w_rows < f.cout || f.kpad < f.cin || g.w_rows < g.cout || g.kpad < g.cin) return 0;
    const long long npix = (long long)f.batch * f.h * f.w_in;
    const long long xb = ((npix - 1) * f.x_ld + f.cin) * 2;
    if (npix <= 0 || npix >= 0x7fffffffLL || xb >= 0x7fffffffLL) return 0;
    P1ChainArgs c{};
    P1Args& a = c.f;
    a.x = static_cast<const f16*>(f.x);
    a.y = keep_first ? static_cast<f16*>(f.y) : nullptr;
    a.w = static_cast<const f16*>(f.w);
    a.bias = f.bias; a.scale = f.scale; a.shift = f.shift;
    a.w_ld = f.kpad; a.x_ld = f.x_ld; a.y_ld = f.y_ld; a.x_bytes = (int)xb;
    a.npix = (int)npix;
    a.slope = f.act == Y5_ACT_LEAKY ? f.slope : 1.f;
    c.w2 = static_cast<const f16*>(g.w);
    c.bias2 = g.bias; c.scale2 = g.scale; c.shift2 = g.shift;
    c.y2 = static_cast<f16*>(g.y);
    c.w2_ld = g.kpad; c.y2_ld = g.y_ld;
    c.slope2 = g.act == Y5_ACT_LEAKY ? g.slope : 1.f;
    *rc = shape == 1 ? launch_chain<16, 8, 8, 4>(c, s) : launch_chain<8, 4, 4, 2>(c, s);
    return 1;
}

// 1x1 stride-1 convolution of [nearest-2x-upsampled x | x2] as the persistent kernel: compiled for 128 + 128 input and 128
// output channels and for 256 + 256 -> 256, without a residual.  Returns 0 when the convolution is outside that (then *rc is untouched), 1 after a
// launch attempt with its status in *rc.  d, x2, x2_ld and cin2 were validated by y5_launch_conv_cat2.
int y5_try_conv_p1x1_cat2(const y5_conv_desc& d, const void* x2, int x2_ld, int cin2, hipStream_t s, int* rc) {
    const bool small = d.cin == 128 && cin2 == 128 && d.cout == 128, big = d.cin == 256 && cin2 == 256 && d.cout == 256;
    if ((!small && !big) || d.res) return 0;
    if (d.w_rows < d.cout || d.kpad < d.cin + cin2) return 0;
    const long long npix = (long long)d.batch * d.ho * d.wo;
    const long long ab = (((long long)d.batch * (d.ho / 2) * (d.wo / 2) - 1) * d.x_ld + d.cin) * 2, bb = ((npix - 1) * x2_ld + cin2) * 2;
    if (npix <= 0 || npix >= 0x7fffffffLL || ab >= 0x7fffffffLL || bb >= 0x7fffffffLL) return 0;
    P1CatArgs c{};
    P1Args& a = c.f;
    a.x = static_cast<const f16*>(d.x);
    a.y = static_cast<f16*>(d.y);
    a.w = static_cast<const f16*>(d.w);
    a.bias = d.bias; a.scale = d.scale; a.shift = d.shift;
    a.w_ld = d.kpad; a.x_ld = d.x_ld; a.y_ld = d.y_ld; a.x_bytes = (int)ab;
    a.npix = (int)npix;
    a.slope = d.act == Y5_ACT_LEAKY ? d.slope : 1.f;
    c.x2 = static_cast<const f16*>(x2);
    c.x2_ld = x2_ld; c.x2_bytes = (int)bb;
    c.H = d.ho; c.W = d.wo;
    c.div_hw = make_fastdiv((unsigned)(d.ho * d.wo));
    c.div_w = make_fastdiv((unsigned)d.wo);
    const int f = (d.bias ? 1 : 0) | (d.scale ? 2 : 0);
    if (small) *rc = f == 0 ? launch_cat_f<8, 4, 1, 0>(c, s) : f == 1 ? launch_cat_f<8, 4, 1, 1>(c, s) : f == 2 ? launch_cat_f<8, 4, 1, 2>(c, s) : launch_cat_f<8, 4, 1, 3>(c, s);
    else *rc = f == 0 ? launch_cat_f<16, 8, 1, 0>(c, s) : f == 1 ? launch_cat_f<16, 8, 1, 1>(c, s) : f == 2 ? launch_cat_f<16, 8, 1, 2>(c, s) : launch_cat_f<16, 8, 1, 3>(c, s);
    return 1;
}
