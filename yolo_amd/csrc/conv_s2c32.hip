// 3x3 stride-2 convolution 32 -> 64 as a PERSISTENT kernel: yolov5s layer 1 (models/yolov5s.yaml backbone, Conv.forward
// models/common.py:26), which reads the largest activation of the graph (32 channels at 320x320) and is bound by memory.
//
// One workgroup per CU walks 8 x 16 output tiles.  A loader wave copies the 17 x 33 input pixels a tile needs global -> LDS
// by DMA, two tiles ahead (three buffers), with the columns DE-INTERLEAVED on the way: odd input columns in one plane, even
// ones in another (interleaved in blocks of eight pixels, so that a copy instruction still reads 16 consecutive pixels), so
// that the three horizontal taps of the stride-2 window read unit-stride rows of LDS instead of every other pixel (which
// would put half the lanes of a ds_read_b128 on the same banks).  Eight compute waves - (two tile
// rows) x (32 output channels) - keep all 3x3 weights of their channels in registers for the whole kernel, multiply the
// nine taps straight from the staged planes and store from registers; one LDS-only barrier per tile, no wait on own stores.
#include <atomic>

#include "y5_common.h"

namespace {

#include "conv_common.h"
#include "epi_mix.h"

struct S2Args {
    const f16* x;
    f16* y;
    const f16* w;
    const float *bias, *scale, *shift;
    int w_ld, H, W, Ho, Wo, x_ld, y_ld, x_bytes;
    int tiles_y, ntiles;
    FastDiv div_tpi, div_ty;
    float slope;
};

constexpr int CI = 32, CO = 64;
constexpr int TH = 8, TW = 16;                              // output tile
constexpr int IR = 2 * TH + 1, IC = 2 * TW + 2, PC = IC / 2;  // staged input rows and columns (-1 .. 32, the last one unused); columns per plane
constexpr int NPX = IR * IC;                                // 578 staged pixels
constexpr int ROW = CI * 2;                                 // 64 bytes per pixel: 4 slots
constexpr int PIECES = (NPX * ROW + 1023) / 1024, X_BYTES = PIECES * 1024, NBUF = 3;
// Two staged layouts of the planes, compiled as two variants (which one is faster is a matter of how the copy's reads
// coalesce against how the fragment reads bank; measured per shape by the caller's autotuner).  Pixel n = row * PC + j of
// plane pl (0: odd input columns, 1: even), 16-byte slot s at s ^ ((n >> 2) & 3) in both:
//   BLK = false: plane after plane, (pl * IR * PC + n) * 64 - a copy instruction reads 16 pixels of one plane, i.e. every
//                other pixel of a 2 KiB stretch;
//   BLK = true:  a 1 KiB copy piece holds 16 CONSECUTIVE input pixels as [8 odd-column pixels][8 even-column pixels]:
//                (n >> 3) * 1024 + pl * 512 + (n & 7) * 64 - whole cache lines per copy instruction.
constexpr int NWC = 8, NTHC = 64 * NWC, NTH = NTHC + 64;
constexpr int PRM0 = NBUF * X_BYTES, SMEM = PRM0 + 3 * CO * 4;
static_assert(SMEM <= 160 * 1024, "fits the CU's LDS");

template <bool BLK>
__device__ __forceinline__ int swz(int pl, int n, int s) {
    const int px = BLK ? (n >> 3) * 1024 + pl * 512 + (n & 7) * ROW : (pl * IR * PC + n) * ROW;
    return px + ((s ^ ((n >> 2) & 3)) << 4);
}

template <int F, bool BLK>
__global__ __launch_bounds__(NTH) void conv_s2c32_kernel(const S2Args a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, lr = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nwg = gridDim.x, g = xcd_remap(blockIdx.x, nwg);
    const int first = (int)((long long)g * a.ntiles / nwg), last = (int)((long long)(g + 1) * a.ntiles / nwg);

    float* prm = reinterpret_cast<float*>(smem + PRM0);
    for (int i = t; i < CO; i += NTH) {
        prm[i] = a.bias ? a.bias[i] : 0.f;
        prm[CO + i] = a.scale ? a.scale[i] : 1.f;
        prm[2 * CO + i] = a.scale ? a.shift[i] : 0.f;
    }

    int tb, ty0, tx0;  // image and first output pixel of a tile
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
        // per-lane constants: byte offset from input pixel (2 ty0 - 1, 2 tx0 - 1), and flags: bit 0 first staged row (above
        // the image for the top tiles), bit 1 first odd column (left of the image for the leftmost tiles), bit 2 nothing to copy
        int rel[PIECES], flag[PIECES];
#pragma unroll
        for (int d = 0; d < PIECES; ++d) {
            int pl, n;
            if constexpr (BLK) {
                pl = lane >> 5;
                n = d * 8 + ((lane & 31) >> 2);
            } else {
                const int p = d * 16 + (lane >> 2);
                pl = p >= IR * PC ? 1 : 0;
                n = p - pl * IR * PC;
            }
            const int s = (lane & 3) ^ ((n >> 2) & 3);
            const int row = n / PC, j = n - row * PC;
            const int col = 2 * j + pl;  // input column relative to 2 tx0 - 1
            rel[d] = (row * a.W + col) * (a.x_ld * 2) + s * 16;
            flag[d] = (row == 0 ? 1 : 0) | (col == 0 ? 2 : 0) | (row >= IR || col == IC - 1 || pl > 1 ? 4 : 0);
        }
        __builtin_amdgcn_s_setprio(3);
        auto issue = [&](int tile, int buf) {
            locate(tile);
            const int base = ((tb * a.H + 2 * ty0 - 1) * a.W + 2 * tx0 - 1) * (a.x_ld * 2);
            const int dead = 4 | (ty0 == 0 ? 1 : 0) | (tx0 == 0 ? 2 : 0);
#pragma unroll
            for (int d = 0; d < PIECES; ++d)
                dma16_buf(xr, (flag[d] & dead) ? Y5_BUF_OOB : (unsigned)(base + rel[d]), (unsigned)(buf * X_BYTES + d * 1024));
        };
        // two tiles ahead: at the barrier of tile i the copies of tile i + 1 may still be in flight (the newest PIECES)
        if (first < last) issue(first, 0);
        if (first + 1 < last) issue(first + 1, 1);
        int buf = 2;
        for (int tile = first; tile < last; ++tile) {
            if (tile + 1 < last)
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PIECES) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (tile + 2 < last) issue(tile + 2, buf);  // this buffer held tile - 1, which every wave has finished
            buf = buf == NBUF - 1 ? 0 : buf + 1;
        }
        return;
    }

    // ---------------------------------------------------------------------- compute waves
    const int nt = wave >> 1, mt = wave & 1;  // output rows 2 nt, 2 nt + 1 of the tile; output channels [32 mt, +32)
    const int r = 2 * nt + (lr >> 4), c = lr & 15;
    f16x8 wf[9 * CI / 16];
#pragma unroll
    for (int ks = 0; ks < 9 * CI / 16; ++ks) wf[ks] = *reinterpret_cast<const f16x8*>(a.w + (size_t)(mt * 32 + perm_row(lr)) * a.w_ld + ks * 16 + lh * 8);
    int uoff[9];  // byte offset of this lane's K-step-0 fragment per tap; the second 16 channels: offset ^ 32
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int row = 2 * r + tap / 3, dx = tap % 3;
        uoff[tap] = dx == 1 ? swz<BLK>(1, row * PC + c, lh) : swz<BLK>(0, row * PC + c + (dx == 2 ? 1 : 0), lh);
    }
    Prm16<F> q;
    float negzero;
    asm volatile("v_mov_b32 %0, 0x80000000" : "=v"(negzero));
    int buf = 0;
    bool have_prm = false;
    for (int tile = first; tile < last; ++tile) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (!have_prm) {  // the parameters were staged by all threads before the first barrier
            q.load(prm, CO, mt * 32 + 16 * lh);
            have_prm = true;
        }
        locate(tile);
        const unsigned char* X = smem + buf * X_BYTES;
        f32x16 acc = zero16();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < CI / 16; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[tap * (CI / 16) + ks], *reinterpret_cast<const f16x8*>(X + (uoff[tap] ^ (ks << 5))), acc, 0, 0, 0);
        f16x8 lo, hi;
        finish<F>(acc, q, a.slope, negzero, lo, hi);
        f16* yp = a.y + ((size_t)(tb * a.Ho + ty0 + r) * a.Wo + tx0 + c) * a.y_ld + mt * 32 + 16 * lh;
        *reinterpret_cast<f16x8*>(yp) = lo;
        *reinterpret_cast<f16x8*>(yp + 8) = hi;
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

template <int F, bool BLK>
int launch(const S2Args& a, hipStream_t s) {
    static std::atomic<unsigned long long> attr_set{0};
    const unsigned long long bit = 1ull << y5_current_device();
    if (!(attr_set.load() & bit)) {
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_s2c32_kernel<F, BLK>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        attr_set.fetch_or(bit);
    }
    const int cus = cu_count();
    hipLaunchKernelGGL((conv_s2c32_kernel<F, BLK>), dim3((unsigned)(a.ntiles < cus ? a.ntiles : cus)), dim3(NTH), (size_t)SMEM, s, a);
    Y5_LAUNCH_CHECK("conv_s2c32_kernel");
    return Y5_OK;
}

}  // namespace

int y5_conv_s2c32_count() { return 2; }

// which = 1 (planes one after the other) or 2 (planes interleaved in blocks of eight pixels).  Returns 0 when the convolution is outside what this kernel is compiled for (then *rc is untouched), 1 after a
// launch attempt with its status in *rc.  d was validated by y5_launch_conv.
int y5_try_conv_s2c32(const y5_conv_desc& d, hipStream_t s, int which, int* rc) {
    if (which < 1 || which > 2 || d.ksize != 3 || d.stride != 2 || d.pad != 1 || d.cin != CI || d.cout != CO || d.res) return 0;
    if (d.h != 2 * d.ho || d.w_in != 2 * d.wo || d.ho % TH || d.wo % TW) return 0;
    if (d.w_rows < CO || d.kpad < 9 * CI) return 0;
    const int f = (d.bias ? 1 : 0) | (d.scale ? 2 : 0);
    const long long xb = (((long long)d.batch * d.h * d.w_in - 1) * d.x_ld + CI) * 2;
    if (xb >= 0x7fffffffLL) return 0;
    S2Args a{};
    a.x = static_cast<const f16*>(d.x);
    a.y = static_cast<f16*>(d.y);
    a.w = static_cast<const f16*>(d.w);
    a.bias = d.bias; a.scale = d.scale; a.shift = d.shift;
    a.w_ld = d.kpad; a.H = d.h; a.W = d.w_in; a.Ho = d.ho; a.Wo = d.wo; a.x_ld = d.x_ld; a.y_ld = d.y_ld; a.x_bytes = (int)xb;
    a.tiles_y = d.ho / TH;
    const int tiles_x = d.wo / TW;
    const long long total = (long long)d.batch * a.tiles_y * tiles_x;
    if (total <= 0 || total > 0x7fffffffLL) return 0;
    a.ntiles = (int)total;
    a.div_tpi = make_fastdiv((unsigned)(a.tiles_y * tiles_x));
    a.div_ty = make_fastdiv((unsigned)a.tiles_y);
    a.slope = d.act == Y5_ACT_LEAKY ? d.slope : 1.f;
    if (which == 1) *rc = f == 0 ? launch<0, false>(a, s) : f == 1 ? launch<1, false>(a, s) : f == 2 ? launch<2, false>(a, s) : launch<3, false>(a, s);
    else *rc = f == 0 ? launch<0, true>(a, s) : f == 1 ? launch<1, true>(a, s) : f == 2 ? launch<2, true>(a, s) : launch<3, true>(a, s);
    return 1;
}
