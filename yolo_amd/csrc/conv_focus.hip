// Focus + its 3x3 convolution in one kernel for gfx950 (models/common.py:79-86 followed by Conv.forward, :26).
//
// The space-to-depth gather followed by a 3x3/pad-1 convolution over 4*3 channels is a 6x6/stride-2/pad-2 convolution
// over the 3-channel image: tap (r, q) on sub-pixel (dy, dx) reads image pixel (2y + 2r+dy - 2, 2x + 2q+dx - 2).  Doing
// it from the NCHW image directly saves writing and re-reading the 12(16)-channel space-to-depth tensor, which costs
// more HBM traffic than the convolution's own output.
//
// A workgroup (4 waves) owns 8x32 output pixels: the 20x70x3 image patch goes to LDS once, each wave multiplies two
// rows of 32 pixels.  K is ordered (r6, c, q6) with q6 padded from 6 to 8 (zero weights): one MFMA k-group of 8 is
// then 8 consecutive fp16 of one image row in the patch, so the pixel operand needs no gather.  The weights are
// re-ordered from the packed (r, q, channel) layout into LDS once per (persistent) workgroup.
#include "y5_common.h"

namespace {

struct FocusArgs {
    const void* img;  // [B][3][H][W] fp16, or uint8 to be divided by 255: [B][3][H][W], or [B][H][W][3] (IN == 2)
    int bgr;          // IN == 2: the three interleaved bytes of a pixel are B, G, R (the model's channel c is byte 2 - c)
    const f16* w;     // packed rows, k = (r*3 + q)*16 + (dy + 2*dx)*3 + c
    const float *bias, *scale, *shift;
    f16* y;
    int B, H, W, Ho, Wo, Cout, y_ld, w_ld;
    int tiles_x, tiles_y, ntiles;
    float slope;
};

// The patch holds 21 rows (20 used) of 36 dwords (pixel pairs; 70 columns used) per channel: 63 rows = 9 slabs of 7
// rows, and 7 rows x 36 dwords = 252 threads, so thread (r7, dw) stages dword (7*i + r7, dw) of slab i = 0..8 -- its
// global offset is the same for every slab up to a wave-uniform term, which keeps the staging free of index arithmetic.
constexpr int kPatchRows = 21, kPatchDw = 36, kSlabRows = 7, kSlabs = 9;
constexpr int kPatchBuf = 3 * kPatchRows * kPatchDw;  // dwords per patch buffer
constexpr int kWPitch = 152;                       // halfs per weight row in LDS: 18 k-groups of 8 + 8 pad (19 slots: conflict-free)

// Post-ops of one wave's two rows of 32 pixels -> the wave's LDS scratch -> 16-byte stores of whole pixels.
template <int RN, bool BIAS, bool SCALE>
__device__ __forceinline__ void finish_tile(const f32x16 (&acc)[2][RN], const FocusArgs& a, const float* prm, unsigned char* escr,
                                            int b, int oy0, int tx0, int lane) {
    constexpr int CN = RN * 32, EROW = CN * 2 + 16, CPW = CN / 8;
    const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int j = 0; j < RN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = j * 32 + 8 * g + 4 * lh;
                f32x4 v = {acc[m][j][4 * g], acc[m][j][4 * g + 1], acc[m][j][4 * g + 2], acc[m][j][4 * g + 3]};
                if (BIAS) v += *reinterpret_cast<const f32x4*>(prm + nl);
                f16x4 h = __builtin_convertvector(v, f16x4);
                if (SCALE)
                    h = __builtin_convertvector(__builtin_convertvector(h, f32x4) * *reinterpret_cast<const f32x4*>(prm + CN + nl) +
                                                    *reinterpret_cast<const f32x4*>(prm + 2 * CN + nl),
                                                f16x4);
                *reinterpret_cast<f16x4*>(escr + lr * EROW + nl * 2) = leaky4_f16(h, a.slope);
            }
        }
        const int oy = oy0 + m;
#pragma unroll
        for (int p = 0; p < 32 * CPW / 64; ++p) {
            const int e = p * 64 + lane;
            const int px = e / CPW, cc = e - px * CPW;
            const int ox = tx0 + px;
            if (oy < a.Ho && ox < a.Wo && cc * 8 < a.Cout)
                *reinterpret_cast<f16x8*>(a.y + (((size_t)b * a.Ho + oy) * a.Wo + ox) * a.y_ld + cc * 8) =
                    *reinterpret_cast<const f16x8*>(escr + px * EROW + cc * 16);
        }
    }
}

// U8: the image is uint8 and the kernel applies the reference's pre-processing img.half() / 255.0 (test.py:152-154) on
// the way in, through a 256-entry table of fp16(float(v) / 255.0f) -- the input then costs a quarter of the fp16 bytes.
// IN == 2: uint8 in the decoder's own layout, [B][H][W][3] (what cv2.imread / the reference's letterbox produce before
// `img[:, :, ::-1].transpose(2, 0, 1)`, utils/datasets.py:140): the BGR -> RGB swap and the HWC -> CHW transpose happen in
// the fetch (two byte loads per channel and pixel pair instead of one 16-bit load), nothing is copied beforehand.
template <int RN, int IN>
__global__ __launch_bounds__(256) void focus_conv_kernel(const FocusArgs a) {
    constexpr bool U8 = IN != 0, HWC = IN == 2;
    constexpr int CN = RN * 32;                    // channels held by the workgroup
    constexpr int EROW = CN * 2 + 16;              // bytes per pixel in a wave's output scratch
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f16* wl = reinterpret_cast<f16*>(smem);                                        // [CN][kWPitch]
    uint32_t* patch = reinterpret_cast<uint32_t*>(smem + CN * kWPitch * 2);        // 2 x [3][kPatchRows][kPatchDw] (+ slack)
    unsigned char* scr = smem + CN * kWPitch * 2 + 2 * kPatchBuf * 4;              // [4 waves][32][EROW]
    float* prm = reinterpret_cast<float*>(scr + 4 * 32 * EROW);                    // [3][CN]
    f16* norm = reinterpret_cast<f16*>(prm + 3 * CN);                               // [256], U8 only

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 31, lh = lane >> 5;

    for (int idx = t; idx < CN * 144; idx += 256) {
        const int n = idx / 144, rem = idx - n * 144;
        const int g = rem >> 3, q6 = rem & 7;
        const int r6 = g / 3, c = g - 3 * r6;
        f16 v = (f16)0.f;
        if (q6 < 6) v = a.w[(size_t)n * a.w_ld + ((r6 >> 1) * 3 + (q6 >> 1)) * 16 + ((r6 & 1) + 2 * (q6 & 1)) * 3 + c];
        wl[n * kWPitch + rem] = v;
    }
    if (U8) norm[t] = (f16)((float)t / 255.0f);  // 256 threads, 256 entries; published by the first barrier below
    if (t < CN) {
        const bool in = t < a.Cout;
        prm[t] = a.bias && in ? a.bias[t] : 0.f;
        prm[CN + t] = a.scale && in ? a.scale[t] : 1.f;
        prm[2 * CN + t] = a.scale && in ? a.shift[t] : 0.f;
    }
    const bool has_bias = a.bias != nullptr, has_scale = a.scale != nullptr;

    // MFMA step s multiplies k-groups g = s (lanes 0-31) and g = s + 9 (lanes 32-63); group g is row r6 = g/3 of
    // channel c = g%3, so the upper half-wave reads 3 patch rows further down: a per-lane constant.
    unsigned char* escr = scr + wave * 32 * EROW;
    const int tiles_per_img = a.tiles_x * a.tiles_y;

    // Patch staging.  Two patch buffers: while a tile is multiplied from one, the next tile's dwords (fetched into
    // registers one tile ahead) are stored to the other, and the fetch after next is issued before the epilogue.
    const int r7 = t / kPatchDw, dwc = t - r7 * kPatchDw;
    const bool stager = t < kSlabRows * kPatchDw;
    const unsigned voff = (unsigned)(r7 * a.W + 2 * dwc) * (HWC ? 3u : U8 ? 1u : 2u);  // bytes from a slab's first element
    uint32_t pre[kSlabs];
    auto fetch = [&](int tile) {
        const int b = tile / tiles_per_img, rem = tile - b * tiles_per_img;
        const int iy0 = 2 * ((rem / a.tiles_x) * 8) - 2, ix0 = 2 * ((rem % a.tiles_x) * 32) - 2;
        constexpr int ESZ = U8 ? 1 : 2;  // bytes per image element
        const unsigned char* origin =
            HWC ? static_cast<const unsigned char*>(a.img) + (((ptrdiff_t)b * a.H + iy0) * a.W + ix0) * 3
                : static_cast<const unsigned char*>(a.img) + ((ptrdiff_t)b * 3 * a.H * a.W + (ptrdiff_t)iy0 * a.W + ix0) * ESZ;
        // first element of slab i (channel i / 3, rows 7 * (i % 3) ...): wave-uniform
        auto slab_of = [&](int i) {
            if constexpr (HWC) return origin + (size_t)(kSlabRows * (i % 3)) * a.W * 3 + (a.bgr ? 2 - i / 3 : i / 3);
            else return origin + ((size_t)(i / 3) * a.H + kSlabRows * (i % 3)) * a.W * ESZ;
        };
        auto load2 = [&](const unsigned char* p) -> uint32_t {  // two neighbouring pixels of one channel
            if constexpr (HWC) return (uint32_t)p[0] | (uint32_t)p[3] << 8;
            else if constexpr (U8) return *reinterpret_cast<const unsigned short*>(p);
            else return *reinterpret_cast<const uint32_t*>(p);
        };
        const bool interior = iy0 >= 0 && iy0 + kPatchRows <= a.H && ix0 >= 0 && ix0 + 2 * kPatchDw <= a.W;
        if (!stager) return;
        if (interior) {
#pragma unroll
            for (int i = 0; i < kSlabs; ++i) pre[i] = load2(slab_of(i) + voff);
        } else {
            const int ix = ix0 + 2 * dwc;
            const bool xok = ix >= 0 && ix < a.W;
#pragma unroll
            for (int i = 0; i < kSlabs; ++i) {
                const int iy = iy0 + kSlabRows * (i % 3) + r7;
                const bool ok = xok && iy >= 0 && iy < a.H;
                const uint32_t v = load2(ok ? slab_of(i) + voff : static_cast<const unsigned char*>(a.img));
                pre[i] = ok ? v : 0u;  // padding: byte 0 maps to norm[0] = 0.0 as well
            }
        }
    };
    auto stage = [&](int buf) {
        if (!stager) return;
#pragma unroll
        for (int i = 0; i < kSlabs; ++i) {
            uint32_t v = pre[i];
            if (U8) {  // two pixels, one byte each -> two normalised halfs (padding stays zero: norm[0] = 0)
                const unsigned short lo = __builtin_bit_cast(unsigned short, norm[v & 255u]);
                const unsigned short hi = __builtin_bit_cast(unsigned short, norm[(v >> 8) & 255u]);
                v = (uint32_t)lo | (uint32_t)hi << 16;
            }
            patch[buf * kPatchBuf + i * kSlabRows * kPatchDw + t] = v;
        }
    };
    const int first = blockIdx.x, stride = gridDim.x;
    if (U8) __syncthreads();  // the normalisation table is read by the very first stage()
    if (first < a.ntiles) {
        fetch(first);
        stage(0);
        if (first + stride < a.ntiles) fetch(first + stride);
    }
    __syncthreads();  // patch 0, wl and prm are written
    int cur = 0;

    for (int tile = first; tile < a.ntiles; tile += stride) {
        const int b = tile / tiles_per_img, rem = tile - b * tiles_per_img;
        const int ty0 = (rem / a.tiles_x) * 8, tx0 = (rem % a.tiles_x) * 32;
        const unsigned char* pbase = reinterpret_cast<const unsigned char*>(patch + cur * kPatchBuf);

        f32x16 acc[2][RN];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int j = 0; j < RN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[m][j][e] = 0.f;
        // pixel (row 2*wave + m, column lr) starts at patch row 2*(2*wave + m), dword lr
        const int lane_off = ((2 * (2 * wave) + 3 * lh) * kPatchDw + lr) * 4;
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            f16x8 xf[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const uint32_t* p = reinterpret_cast<const uint32_t*>(pbase + lane_off + (((s % 3) * kPatchRows + s / 3 + 2 * m) * kPatchDw) * 4);
                const uint32_t d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3];
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 u = {d0, d1, d2, d3};
                xf[m] = __builtin_bit_cast(f16x8, u);
            }
#pragma unroll
            for (int j = 0; j < RN; ++j) {
                const f16x8 wf = *reinterpret_cast<const f16x8*>(wl + (j * 32 + lr) * kWPitch + (s + 9 * lh) * 8);
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, xf[m], acc[m][j], 0, 0, 0);
            }
        }

        if (tile + stride < a.ntiles) {
            stage(cur ^ 1);  // last read by the previous tile's products, which every wave finished before the barrier below
            if (tile + 2 * stride < a.ntiles) fetch(tile + 2 * stride);
        }

        // The post-op mode is wave-uniform: branch once per tile into a straight-line epilogue.
        if (has_bias && !has_scale) finish_tile<RN, true, false>(acc, a, prm, escr, b, ty0 + 2 * wave, tx0, lane);
        else if (!has_bias && has_scale) finish_tile<RN, false, true>(acc, a, prm, escr, b, ty0 + 2 * wave, tx0, lane);
        else if (has_bias) finish_tile<RN, true, true>(acc, a, prm, escr, b, ty0 + 2 * wave, tx0, lane);
        else finish_tile<RN, false, false>(acc, a, prm, escr, b, ty0 + 2 * wave, tx0, lane);
        __syncthreads();
        cur ^= 1;
    }
}

template <int RN, int IN>
int launch_focus(const FocusArgs& a, hipStream_t s) {
    constexpr int CN = RN * 32;
    constexpr size_t lds = (size_t)CN * kWPitch * 2 + 2 * kPatchBuf * 4 + 4 * 32 * (CN * 2 + 16) + 3 * CN * 4 + 256 * 2;
    static std::atomic<unsigned long long> attr_set{0};  // one bit per device: the attribute is per function and device
    const unsigned long long attr_set_bit = 1ull << y5_current_device();
    if (!(attr_set.load() & attr_set_bit)) {
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(focus_conv_kernel<RN, IN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set.fetch_or(attr_set_bit);
    }
    const int per_cu = RN == 1 ? 4 : RN == 2 ? 3 : 2;
    const int grid = a.ntiles < 256 * per_cu ? a.ntiles : 256 * per_cu;
    hipLaunchKernelGGL((focus_conv_kernel<RN, IN>), dim3((unsigned)grid), dim3(256), lds, s, a);
    Y5_LAUNCH_CHECK("focus_conv_kernel");
    return Y5_OK;
}


// ------------------------------------------------------------------------------------------------------------------
// First-layer 3x3 stride-1 convolution straight from the 3-channel NCHW image (the stem of the graphs without Focus,
// models/yolov5s_ori.yaml layer 0): no packed NHWC copy of the image, whose 8-channel padding alone is 2.7x the image.
// A workgroup (4 waves) owns 8 rows x 64 columns: the 10x66x3 patch goes to LDS once, each wave multiplies two rows.
// K is ordered (c, r, q4) with q4 = the three taps of a row padded to four (zero weight): a k-group of 8 is then two
// runs of 4 consecutive fp16 of patch rows, fetched as dwords and shifted by one halfword for odd columns.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kStemPW = 72;                    // halfs per patch row: 34 dwords used (image columns tx0-2 .. tx0+65), 2 zero
constexpr int kStemPatch = 3 * 10 * kStemPW;   // halfs per patch buffer
constexpr int kStemDw = 3 * 10 * 34;           // dwords staged per tile
constexpr int kStemWP = 56;                    // halfs per weight row in LDS: 6 k-groups of 8 + 8 pad

template <int RN, bool U8>
__global__ __launch_bounds__(256) void stem_conv_kernel(const FocusArgs a) {
    constexpr int CN = RN * 32;
    constexpr int EROW = CN * 2 + 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f16* wl = reinterpret_cast<f16*>(smem);                                   // [CN][kStemWP]
    uint32_t* patch = reinterpret_cast<uint32_t*>(smem + CN * kStemWP * 2);   // 2 x [3][10][kStemPW / 2]
    unsigned char* scr = smem + CN * kStemWP * 2 + 2 * kStemPatch * 2;        // [4 waves][32][EROW]
    float* prm = reinterpret_cast<float*>(scr + 4 * 32 * EROW);               // [3][CN]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 31, lh = lane >> 5;

    for (int idx = t; idx < CN * 48; idx += 256) {
        const int n = idx / 48, kk = idx - n * 48;
        const int pr = 2 * (kk >> 3) + ((kk & 7) >> 2), q = kk & 3;  // pr = c*3 + r
        const int c = pr / 3, r = pr - 3 * c;
        f16 v = (f16)0.f;
        if (pr < 9 && q < 3 && n < a.Cout) v = a.w[(size_t)n * a.w_ld + (r * 3 + q) * 8 + c];  // packed for 8 input channels
        wl[n * kStemWP + kk] = v;
    }
    for (int i = t; i < kStemPatch; i += 256) patch[i] = 0u;  // both buffers (2 x kStemPatch halfs = kStemPatch dwords)
    if (t < CN) {
        const bool in = t < a.Cout;
        prm[t] = a.bias && in ? a.bias[t] : 0.f;
        prm[CN + t] = a.scale && in ? a.scale[t] : 1.f;
        prm[2 * CN + t] = a.scale && in ? a.shift[t] : 0.f;
    }
    const bool has_bias = a.bias != nullptr, has_scale = a.scale != nullptr;
    unsigned char* escr = scr + wave * 32 * EROW;
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    // pixel column x reads patch columns x+1 .. x+3 (patch column 0 is image column tx0-2): even x starts in the upper
    // half of a dword
    const unsigned sh = (unsigned)((lr & 1) ^ 1) * 16u;

    // Patch staging, one tile ahead in registers: thread t owns dwords t, t+256, t+512, t+768 of the 1020 per tile.
    int slot[4];      // dword index in a patch buffer, -1 beyond the tile
    int srow[4], sdw[4], sch[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = t + 256 * i;
        const int c = idx / 340, r2 = idx - c * 340;
        const int r = r2 / 34, dwc = r2 - r * 34;
        sch[i] = c; srow[i] = r; sdw[i] = dwc;
        slot[i] = idx < kStemDw ? (c * 10 + r) * (kStemPW / 2) + dwc : -1;
    }
    uint32_t pre[4];
    auto fetch = [&](int tile) {
        const int b = tile / tiles_per_img, rem = tile - b * tiles_per_img;
        const int ty0 = (rem / a.tiles_x) * 8, tx0 = (rem % a.tiles_x) * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int iy = ty0 - 1 + srow[i], ix = tx0 - 2 + 2 * sdw[i];  // W is even: a dword is inside or outside as a whole
            uint32_t v = 0u;
            if (slot[i] >= 0 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
                const size_t off = ((size_t)(b * 3 + sch[i]) * a.H + iy) * a.W + ix;
                if (U8) {  // two pixels, one byte each: img.half() / 255.0
                    const unsigned short two = *reinterpret_cast<const unsigned short*>(static_cast<const unsigned char*>(a.img) + off);
                    const f16 lo = (f16)((float)(two & 255u) / 255.0f), hi = (f16)((float)(two >> 8) / 255.0f);
                    v = (uint32_t)__builtin_bit_cast(unsigned short, lo) | (uint32_t)__builtin_bit_cast(unsigned short, hi) << 16;
                } else {
                    v = *reinterpret_cast<const uint32_t*>(static_cast<const f16*>(a.img) + off);
                }
            }
            pre[i] = v;
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (slot[i] >= 0) patch[buf * (kStemPatch / 2) + slot[i]] = pre[i];
    };
    const int first = blockIdx.x, stride = gridDim.x;
    __syncthreads();  // zeroed patch buffers before the first stage
    if (first < a.ntiles) {
        fetch(first);
        stage(0);
        if (first + stride < a.ntiles) fetch(first + stride);
    }
    __syncthreads();
    int cur = 0;

    for (int tile = first; tile < a.ntiles; tile += stride) {
        const int b = tile / tiles_per_img, rem = tile - b * tiles_per_img;
        const int ty0 = (rem / a.tiles_x) * 8, tx0 = (rem % a.tiles_x) * 64;
        const uint32_t* p32 = patch + cur * (kStemPatch / 2);

        f32x16 acc[2][2][RN];  // [column half][row][channel tile]
#pragma unroll
        for (int xg = 0; xg < 2; ++xg)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int j = 0; j < RN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[xg][m][j][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const int g = 2 * ks + lh;  // k-group of this half-wave
            f16x8 wf[RN];
#pragma unroll
            for (int j = 0; j < RN; ++j) wf[j] = *reinterpret_cast<const f16x8*>(wl + (j * 32 + lr) * kStemWP + g * 8);
#pragma unroll
            for (int xg = 0; xg < 2; ++xg) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    u32x4 u;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        int pr = 2 * g + h;
                        if (pr > 8) pr = 8;  // zero-weight groups: any valid address
                        const int c = pr / 3, r = pr - 3 * c;
                        const int hw = (c * 10 + 2 * wave + m + r) * kStemPW + 32 * xg + lr + 1;  // halfword of tap q = 0
                        const uint32_t d0 = p32[hw >> 1], d1 = p32[(hw >> 1) + 1], d2 = p32[(hw >> 1) + 2];
                        u[2 * h] = __builtin_amdgcn_alignbit(d1, d0, sh);
                        u[2 * h + 1] = __builtin_amdgcn_alignbit(d2, d1, sh);
                    }
                    const f16x8 xf = __builtin_bit_cast(f16x8, u);
#pragma unroll
                    for (int j = 0; j < RN; ++j) acc[xg][m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j], xf, acc[xg][m][j], 0, 0, 0);
                }
            }
        }
        if (tile + stride < a.ntiles) {
            stage(cur ^ 1);  // last read by the previous tile's products, which every wave finished before the barrier below
            if (tile + 2 * stride < a.ntiles) fetch(tile + 2 * stride);
        }
#pragma unroll
        for (int xg = 0; xg < 2; ++xg) {
            if (has_bias && !has_scale) finish_tile<RN, true, false>(acc[xg], a, prm, escr, b, ty0 + 2 * wave, tx0 + 32 * xg, lane);
            else if (!has_bias && has_scale) finish_tile<RN, false, true>(acc[xg], a, prm, escr, b, ty0 + 2 * wave, tx0 + 32 * xg, lane);
            else if (has_bias) finish_tile<RN, true, true>(acc[xg], a, prm, escr, b, ty0 + 2 * wave, tx0 + 32 * xg, lane);
            else finish_tile<RN, false, false>(acc[xg], a, prm, escr, b, ty0 + 2 * wave, tx0 + 32 * xg, lane);
        }
        __syncthreads();
        cur ^= 1;
    }
}

template <int RN, bool U8>
int launch_stem(const FocusArgs& a, hipStream_t s) {
    constexpr int CN = RN * 32;
    constexpr size_t lds = (size_t)CN * kStemWP * 2 + 2 * kStemPatch * 2 + 4 * 32 * (CN * 2 + 16) + 3 * CN * 4;
    const int grid = a.ntiles < 256 * 6 ? a.ntiles : 256 * 6;
    hipLaunchKernelGGL((stem_conv_kernel<RN, U8>), dim3((unsigned)grid), dim3(256), lds, s, a);
    Y5_LAUNCH_CHECK("stem_conv_kernel");
    return Y5_OK;
}

}  // namespace

int y5_launch_focus_conv(const y5_conv_desc& d, hipStream_t s, int input) {
    const bool u8 = input != Y5_IN_F16;
    Y5_REQUIRE(input >= Y5_IN_F16 && input <= Y5_IN_U8_HWC_BGR, "focus_conv: unknown input layout %d", input);
    Y5_REQUIRE(d.dtype == Y5_F16, "focus_conv: fp16 only (dtype %d)", d.dtype);
    Y5_REQUIRE(d.x && d.w && d.y, "focus_conv: null image, weight or output pointer");
    Y5_REQUIRE(d.cin == 3, "focus_conv: the image must have 3 channels (got %d)", d.cin);
    Y5_REQUIRE(d.ksize == 3 && d.stride == 1 && d.pad == 1, "focus_conv: only the 3x3 stride-1 pad-1 convolution is fused");
    Y5_REQUIRE(d.batch > 0 && d.h > 0 && d.w_in > 0 && d.h % 2 == 0 && d.w_in % 2 == 0, "focus_conv: image %dx%dx%d must be even-sized",
               d.batch, d.h, d.w_in);
    Y5_REQUIRE(d.ho == d.h / 2 && d.wo == d.w_in / 2, "focus_conv: output must be %dx%d", d.h / 2, d.w_in / 2);
    Y5_REQUIRE(d.cout > 0 && d.cout <= 96 && d.cout % 8 == 0, "focus_conv: cout %d must be a multiple of 8 up to 96", d.cout);
    Y5_REQUIRE(d.y_ld >= d.cout && d.y_ld % 8 == 0, "focus_conv: y_ld %d must be a multiple of 8 >= cout", d.y_ld);
    Y5_REQUIRE(d.kpad >= 144 && d.w_rows >= 96, "focus_conv: weights must be packed for 16 input channels (kpad >= 144, rows >= 96)");
    Y5_REQUIRE(!d.res, "focus_conv: no residual input");
    Y5_REQUIRE((d.scale == nullptr) == (d.shift == nullptr), "focus_conv: scale and shift come together");
    Y5_REQUIRE(d.act == Y5_ACT_NONE || (d.act == Y5_ACT_LEAKY && d.slope > 0.f && d.slope <= 1.f), "focus_conv: bad activation");
    Y5_REQUIRE((reinterpret_cast<uintptr_t>(d.x) & (input >= Y5_IN_U8_HWC ? 0u : u8 ? 1u : 3u)) == 0 && y5_aligned16(d.y) && y5_aligned16(d.w), "focus_conv: misaligned pointer");
    FocusArgs a{};
    a.img = d.x;
    a.w = static_cast<const f16*>(d.w);
    a.bias = d.bias; a.scale = d.scale; a.shift = d.shift;
    a.y = static_cast<f16*>(d.y);
    a.B = d.batch; a.H = d.h; a.W = d.w_in; a.Ho = d.ho; a.Wo = d.wo; a.Cout = d.cout; a.y_ld = d.y_ld; a.w_ld = d.kpad;
    a.tiles_x = (d.wo + 31) / 32; a.tiles_y = (d.ho + 7) / 8;
    const long long nt = (long long)a.tiles_x * a.tiles_y * d.batch;
    Y5_REQUIRE(nt < (1LL << 31), "focus_conv: too many tiles");
    a.ntiles = (int)nt;
    a.slope = d.act == Y5_ACT_LEAKY ? d.slope : 1.f;
    a.bgr = input == Y5_IN_U8_HWC_BGR;
    if (input >= Y5_IN_U8_HWC) {
        if (d.cout <= 32) return launch_focus<1, 2>(a, s);
        if (d.cout <= 64) return launch_focus<2, 2>(a, s);
        return launch_focus<3, 2>(a, s);
    }
    if (u8) {
        if (d.cout <= 32) return launch_focus<1, 1>(a, s);
        if (d.cout <= 64) return launch_focus<2, 1>(a, s);
        return launch_focus<3, 1>(a, s);
    }
    if (d.cout <= 32) return launch_focus<1, 0>(a, s);
    if (d.cout <= 64) return launch_focus<2, 0>(a, s);
    return launch_focus<3, 0>(a, s);
}

int y5_launch_stem_conv(const y5_conv_desc& d, hipStream_t s, bool u8) {
    Y5_REQUIRE(d.dtype == Y5_F16, "stem_conv: fp16 only (dtype %d)", d.dtype);
    Y5_REQUIRE(d.x && d.w && d.y, "stem_conv: null image, weight or output pointer");
    Y5_REQUIRE(d.cin == 3, "stem_conv: the image must have 3 channels (got %d)", d.cin);
    Y5_REQUIRE(d.ksize == 3 && d.stride == 1 && d.pad == 1, "stem_conv: 3x3 stride-1 pad-1 only");
    Y5_REQUIRE(d.batch > 0 && d.h > 0 && d.w_in > 0 && d.ho == d.h && d.wo == d.w_in, "stem_conv: output must be %dx%d", d.h, d.w_in);
    Y5_REQUIRE(d.w_in % 2 == 0, "stem_conv: the image width %d must be even (pixel pairs are staged)", d.w_in);
    Y5_REQUIRE((long long)d.batch * 3 * d.h * d.w_in < 0x7fffffffLL, "stem_conv: image batch too large");
    Y5_REQUIRE(d.cout > 0 && d.cout <= 64 && d.cout % 8 == 0, "stem_conv: cout %d must be a multiple of 8 up to 64", d.cout);
    Y5_REQUIRE(d.y_ld >= d.cout && d.y_ld % 8 == 0, "stem_conv: y_ld %d must be a multiple of 8 >= cout", d.y_ld);
    Y5_REQUIRE(d.kpad >= 72 && d.w_rows >= d.cout, "stem_conv: weights must be packed for 8 input channels (kpad >= 72)");
    Y5_REQUIRE(!d.res, "stem_conv: no residual input");
    Y5_REQUIRE((d.scale == nullptr) == (d.shift == nullptr), "stem_conv: scale and shift come together");
    Y5_REQUIRE(d.act == Y5_ACT_NONE || (d.act == Y5_ACT_LEAKY && d.slope > 0.f && d.slope <= 1.f), "stem_conv: bad activation");
    Y5_REQUIRE((reinterpret_cast<uintptr_t>(d.x) & (u8 ? 1u : 3u)) == 0 && y5_aligned16(d.y) && y5_aligned16(d.w), "stem_conv: misaligned pointer");
    FocusArgs a{};
    a.img = d.x;
    a.w = static_cast<const f16*>(d.w);
    a.bias = d.bias; a.scale = d.scale; a.shift = d.shift;
    a.y = static_cast<f16*>(d.y);
    a.B = d.batch; a.H = d.h; a.W = d.w_in; a.Ho = d.ho; a.Wo = d.wo; a.Cout = d.cout; a.y_ld = d.y_ld; a.w_ld = d.kpad;
    a.tiles_x = (d.wo + 63) / 64; a.tiles_y = (d.ho + 7) / 8;
    const long long nt = (long long)a.tiles_x * a.tiles_y * d.batch;
    Y5_REQUIRE(nt < (1LL << 31), "stem_conv: too many tiles");
    a.ntiles = (int)nt;
    a.slope = d.act == Y5_ACT_LEAKY ? d.slope : 1.f;
    if (u8) return d.cout <= 32 ? launch_stem<1, true>(a, s) : launch_stem<2, true>(a, s);
    return d.cout <= 32 ? launch_stem<1, false>(a, s) : launch_stem<2, false>(a, s);
}
