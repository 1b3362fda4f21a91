// One Detect level - the head's 1x1 convolution (bias, no activation) with the box decode as its epilogue - as a
// PERSISTENT kernel for heads of 128 or 256 input channels (models/yolo.py:24-46; the finest pyramid level, where
// most of Detect's output is written, has 128 in yolov5s).
//
// The GEMM of a level is tiny per pixel (K <= 256, 255 outputs) and its traffic is nearly all output (x_i and z, 2 x 255
// halfs per pixel), so the kernel is organised around keeping stores in flight: one workgroup per CU walks a contiguous run
// of 64-pixel tiles; each of its 8 compute waves owns 32 output channels and keeps their weights in registers for the
// whole kernel (no weight traffic after the first tile); a ninth wave copies the next tile's input pixels global -> LDS
// by DMA while the current tile is decoded and stored (128 channels; with 256 the weights need more registers than
// a nine-wave workgroup leaves a wave, so there the eight compute waves share the copy, four 1 KiB pieces each, and
// wait for them right before their own copy out, half a tile after issuing them).  The decode runs in the GEMM's epilogue on the accumulator
// registers (every lane busy, no index arithmetic per element), writes raw and decoded values into two LDS images laid
// out like the output runs, and the copy out is 16-byte pieces.  The waves never wait for their own global stores:
// barriers are LDS-only, and a vmcnt wait only ever covers input copies.
#include <atomic>

#include "y5_common.h"

namespace {

#include "conv_common.h"
#include "decode_tile.h"

struct DetArgs {
    const f16* x;
    const f16* w;
    const float* bias;
    int x_ld, x_bytes, w_ld, cout;
    int npix, ntiles, tiles_per_image;
    FastDiv div_tpi, div_nx, div_no;
};

constexpr int NWC = 8, NTHC = 64 * NWC;  // compute waves (threads 0..511); with a loader wave 64 more threads

template <int KS>  // the head has 16 * KS input channels
struct DetCfg {
    static constexpr int SLOTS = 2 * KS;            // 16-byte slots per input pixel; slot s of pixel p is stored at s ^ (p & 15)
    static_assert(SLOTS >= 16 && (SLOTS & (SLOTS - 1)) == 0, "the swizzle needs rows of 16 slots or more");
    static constexpr int XROW = SLOTS * 16, X_BYTES = TP * XROW, PIECES = X_BYTES / 1024;
    // the staged outputs: raw values and decoded values, each [anchor][pixel][slot] packed exactly like the tile's runs of
    // x_i and z in global memory, so that the copy out is whole 16-byte pieces
    static constexpr int DENSE = TP * 256 * 2;
    static constexpr int RAW0 = X_BYTES, DEC0 = RAW0 + DENSE;
    static constexpr int SMEM = DEC0 + DENSE;
    static_assert(SMEM <= 160 * 1024, "LDS layout");
};

__device__ __forceinline__ int perm_row(int i) { return 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3); }  // see conv_csp.hip

// LW: a ninth wave issues the input copies.  KSR <= KS: the head's real width (16 KSR input channels) inside staged rows laid
// out for 16 KS (the swizzle needs a power-of-two number of slots per pixel): slots past it are never fetched - their lanes of
// a copy ask for an out-of-range address - and K-steps past it are not multiplied.
template <int KS, bool LW, int KSR = KS>
__global__ __launch_bounds__(NTHC + (LW ? 64 : 0)) void detect_p_kernel(const DetArgs a, const DecodeArgs dec) {
    typedef DetCfg<KS> Cfg;
    static_assert(KSR >= 1 && KSR <= KS, "real width inside the layout");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, lr = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nwg = gridDim.x, g = xcd_remap(blockIdx.x, nwg);
    const int first = (int)((long long)g * a.ntiles / nwg), last = (int)((long long)(g + 1) * a.ntiles / nwg);

    // copies of a tile's input pixels: pieces d0, d0 + dstep, ... (all of them in the loader wave, every eighth otherwise)
    const i32x4 xr = make_buf(a.x, a.x_bytes);
    auto issue = [&](int tile, int d0, int dstep) {
        const int b = (int)fdiv((unsigned)tile, a.div_tpi);
        const int pix0 = (tile - b * a.tiles_per_image) * TP;
        const int np = min(TP, a.npix - pix0);
        const unsigned base = (unsigned)((b * a.npix + pix0) * a.x_ld * 2);
#pragma unroll
        for (int j = 0; j < Cfg::PIECES; ++j) {
            const int d = d0 + j * dstep;
            if (d >= Cfg::PIECES) break;
            const int slot = d * 64 + lane, px = slot / Cfg::SLOTS, sp = slot % Cfg::SLOTS;
            const int s = sp ^ (px & 15);
            dma16_buf(xr, px < np && (KSR == KS || s < 2 * KSR) ? base + (unsigned)(px * a.x_ld * 2 + s * 16) : Y5_BUF_OOB, (unsigned)(d * 1024));
        }
    };
    if (LW && wave == NWC) {
        // ------------------------------------------------------------------ loader wave
        if (first < last) issue(first, 0, 1);
        for (int tile = first; tile < last; ++tile) {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // A: the tile's pixels have landed
            asm volatile("s_barrier" ::: "memory");                        // B: the products have read them
            if (tile + 1 < last) issue(tile + 1, 0, 1);
            asm volatile("s_barrier" ::: "memory");                        // C
            asm volatile("s_barrier" ::: "memory");                        // D
        }
        return;
    }

    // ---------------------------------------------------------------------- compute waves: output channels [32 wave, +32)
    f16x8 wf[KSR];
#pragma unroll
    for (int ks = 0; ks < KSR; ++ks) wf[ks] = *reinterpret_cast<const f16x8*>(a.w + (size_t)(wave * 32 + perm_row(lr)) * a.w_ld + ks * 16 + lh * 8);
    const int cbase = wave * 32 + 16 * lh;  // the lane's 16 channels: slots o0.. of anchor an0, then slots 0.. of anchor an0 + 1
    f32x4 bias[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = cbase + 4 * q + e;
            bias[q][e] = a.bias && n < a.cout ? a.bias[n] : 0.f;
        }
    const int no = dec.no, nch = dec.na * no;
    const int an0 = (int)fdiv((unsigned)cbase, a.div_no), o0 = cbase - an0 * no;
    f16* raws = reinterpret_cast<f16*>(smem + Cfg::RAW0);
    f16* decs = reinterpret_cast<f16*>(smem + Cfg::DEC0);
    f16* const x_out = dec.x_out;
    f16* const z_out = dec.z_out;
    if (!LW) {
        if (first < last) issue(first, wave, NWC);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the first tile
    }
    for (int tile = first; tile < last; ++tile) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // A: input landed; the previous tile's copy out is done
        const int b = (int)fdiv((unsigned)tile, a.div_tpi);
        const int pix0 = (tile - b * a.tiles_per_image) * TP;
        const int np = min(TP, a.npix - pix0);
        // One 32-pixel half of the tile at a time (16 accumulator registers live): products, then the raw head values
        // fp16(acc + bias) (what Detect.forward receives from its nn.Conv2d) and their decode, element by element in
        // registers, written where the tile's output runs want them.
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int px = nt * 32 + lr;
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KSR; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks], *reinterpret_cast<const f16x8*>(smem + px * Cfg::XROW + (((2 * ks + lh) ^ (px & 15)) << 4)), acc, 0, 0, 0);
            if (nt == 1) {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // B: the input may be overwritten
                if (!LW && tile + 1 < last) issue(tile + 1, wave, NWC);
            }
            const int row = px * no;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                const f16x4 h = __builtin_convertvector(v + bias[q], f16x4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * q + e;
                    if (cbase + i >= nch) continue;
                    const bool nxt = o0 + i >= no;
                    const int o = nxt ? o0 + i - no : o0 + i;
                    const int off = (nxt ? an0 + 1 : an0) * (TP * no) + row + o;
                    raws[off] = h[e];
                    decs[off] = sigmoid_h<f16>(h[e]);  // the four box slots per anchor are redone below
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // C: both staged tiles are complete
        // the box slots (o < 4) of every (pixel, anchor): decoded from the staged raw values, one per thread
        for (int i = t; i < np * dec.na * 4; i += NTHC) {
            const int o = i & 3, pa = i >> 2;
            const int an = pa / np, p = pa - an * np;
            const int pix = pix0 + p;
            const int gy = (int)fdiv((unsigned)pix, a.div_nx), gx = pix - gy * dec.nx;
            const int off = an * (TP * no) + p * no + o;
            decs[off] = decode_one<f16>(raws[off], o, (float)gx, (float)gy, dec.anchor_wh[2 * an], dec.anchor_wh[2 * an + 1], dec.stride);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // D
        if (!LW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next tile's pieces, issued after B; older than any store still in flight
        // copy out: per anchor np * no consecutive halfs of x_i and of z, starting on 16-byte boundaries
        const int run = np * no, nwin = run >> 3;
        const int npix_img = a.npix;
        for (int an = 0; an < dec.na; ++an) {
            const long long xbase = (((long long)b * dec.na + an) * npix_img + pix0) * no;
            const long long zbase = ((long long)b * dec.z_rows_total + dec.z_row0 + (long long)an * npix_img + pix0) * no;
            const f16* rs = raws + an * (TP * no);
            const f16* ds = decs + an * (TP * no);
            for (int wi = t; wi < nwin; wi += NTHC) {
                if (x_out) *reinterpret_cast<f16x8*>(x_out + xbase + wi * 8) = *reinterpret_cast<const f16x8*>(rs + wi * 8);
                *reinterpret_cast<f16x8*>(z_out + zbase + wi * 8) = *reinterpret_cast<const f16x8*>(ds + wi * 8);
            }
            for (int e = nwin * 8 + t; e < run; e += NTHC) {  // a last tile whose run is not whole pieces
                if (x_out) x_out[xbase + e] = rs[e];
                z_out[zbase + e] = ds[e];
            }
        }
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

template <int KS, bool LW, int KSR = KS>
int launch(const DetArgs& a, const DecodeArgs& dec, hipStream_t s) {
    typedef DetCfg<KS> Cfg;
    static std::atomic<unsigned long long> attr_set{0};
    const unsigned long long bit = 1ull << y5_current_device();
    if (!(attr_set.load() & bit)) {
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(detect_p_kernel<KS, LW, KSR>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
        attr_set.fetch_or(bit);
    }
    const int cus = cu_count();
    hipLaunchKernelGGL((detect_p_kernel<KS, LW, KSR>), dim3((unsigned)(a.ntiles < cus ? a.ntiles : cus)), dim3(NTHC + (LW ? 64 : 0)), (size_t)Cfg::SMEM, s, a, dec);
    Y5_LAUNCH_CHECK("detect_p_kernel");
    return Y5_OK;
}

}  // namespace

// The arguments were validated by y5_launch_detect.  Returns 0 when the level is outside what this kernel is compiled for
// (then *rc is untouched), 1 after a launch attempt with its status in *rc.
int y5_try_detect_persistent(const y5_conv_desc& c, const y5_decode_desc& d, hipStream_t s, int* rc) {
    // 128 and 256 input channels, and the widths of yolov5m / yolov5x heads inside those layouts' next power of two (192 in rows
    // of 256, 320 and 384 in rows of 512); 512 channels would take 128 weight registers per lane and spill
    if (c.cin != 128 && c.cin != 256 && c.cin != 192 && c.cin != 320 && c.cin != 384) return 0;
    DecodeArgs dec;
    size_t lds_bytes = 0;
    bool vec = false;
    if (prepare_level(d, dec, &lds_bytes, &vec) != Y5_OK || !vec || d.na * d.no > 256) return 0;
    const long long xb = (((long long)c.batch * c.h * c.w_in - 1) * c.x_ld + c.cin) * 2;
    if (xb >= 0x7fffffffLL) return 0;
    DetArgs a{};
    a.x = static_cast<const f16*>(c.x);
    a.w = static_cast<const f16*>(c.w);
    a.bias = c.bias;
    a.x_ld = c.x_ld; a.x_bytes = (int)xb; a.w_ld = c.kpad; a.cout = c.cout;
    a.npix = d.ny * d.nx;
    a.tiles_per_image = (a.npix + TP - 1) / TP;
    const long long total = (long long)a.tiles_per_image * c.batch;
    if (total <= 0 || total > 0x7fffffffLL) return 0;
    a.ntiles = (int)total;
    a.div_tpi = make_fastdiv((unsigned)a.tiles_per_image);
    a.div_nx = make_fastdiv((unsigned)d.nx);
    a.div_no = make_fastdiv((unsigned)d.no);
    if ((TP * d.no) % 8) return 0;  // per-anchor blocks of the staged tiles start on 16-byte boundaries
    if (d.no < 16) return 0;        // a lane's 16 channels cross at most one anchor boundary
    *rc = c.cin == 128 ? launch<8, true>(a, dec, s) : c.cin == 256 ? launch<16, false>(a, dec, s) : c.cin == 192 ? launch<16, false, 12>(a, dec, s)
        : c.cin == 320 ? launch<32, false, 20>(a, dec, s) : launch<32, false, 24>(a, dec, s);
    return 1;
}
