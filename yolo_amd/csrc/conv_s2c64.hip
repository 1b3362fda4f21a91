// 3x3 stride-2 convolution 64 -> 128 as a PERSISTENT kernel: yolov5s layer 3 (models/yolov5s.yaml backbone, Conv.forward
// models/common.py:26).  K = 576 is long enough for the matrix pipes to be the bound once neither operand streams through
// LDS rings: the weights never touch LDS at all.
//
// One workgroup per CU walks 4 x 16 output tiles.  Eight waves - (two tile rows) x (32 output channels) - keep all 3x3
// weights of their channels in registers for the whole kernel (36 fragments = 144 registers: eight waves of 256 registers,
// so there is no loader wave).  Every wave copies five 1 KiB pieces of the 9 x 34 input pixels a tile needs global -> LDS
// by DMA, two tiles ahead (three buffers), with the columns DE-INTERLEAVED on the way (odd input columns in one plane, even
// ones in another) so that the three horizontal taps of the stride-2 window read unit-stride rows of LDS; the 16-byte slots
// of a pixel are swizzled by its plane column, which makes every ds_read_b128 pass (eight pixels of one plane row, eight of
// the row two below, shifted by the tap) conflict-free.  The waves multiply the nine taps straight from the planes and
// store from registers: one LDS-only barrier per tile, and the one vmcnt wait per tile - for the copies issued a whole tile
// earlier, with the copies issued at the top of this tile allowed to stay in flight - sits right before the tile's stores.
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

constexpr int CI = 64, CO = 128;
constexpr int TH = 4, TW = 16;                                // output tile
constexpr int IR = 2 * TH + 1, IC = 2 * TW + 2, PC = IC / 2;  // staged input rows and columns (-1 .. 32, the last one unused); columns per plane (17)
constexpr int ROW = CI * 2;                                   // 128 bytes per pixel: 8 slots
constexpr int NWC = 8, NTH = 64 * NWC;
constexpr int MYP = 5, PIECES = MYP * NWC;                    // 1 KiB copy pieces per wave and per tile (the planes fill 38.25 of the 40)
constexpr int X_BYTES = PIECES * 1024, NBUF = 3;
static_assert(2 * IR * PC * ROW <= X_BYTES, "the planes fit the pieces");
constexpr int PRM0 = NBUF * X_BYTES, SMEM = PRM0 + 3 * CO * 4;
static_assert(SMEM <= 160 * 1024, "fits the CU's LDS");

// 16-byte slot s of pixel (row, j) of plane pl (0: odd input columns, i.e. columns 2 tx0 - 1, + 1, ...; 1: even)
__device__ __forceinline__ int swz(int pl, int row, int j, int s) { return ((pl * IR + row) * PC + j) * ROW + ((s ^ ((j >> 1) & 7)) << 4); }

template <int F>
__global__ __launch_bounds__(NTH) void conv_s2c64_kernel(const S2Args a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, lr = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nwg = gridDim.x, g = xcd_remap(blockIdx.x, nwg);
    const int first = (int)((long long)g * a.ntiles / nwg), last = (int)((long long)(g + 1) * a.ntiles / nwg);

    float* prm = reinterpret_cast<float*>(smem + PRM0);
    static_assert(CO <= NTH, "one channel's parameters per thread");
    float pv[3] = {0.f, 1.f, 0.f};  // fetched here, stored once the weights have landed: one round trip for everything
    if (t < CO) {
        if (a.bias) pv[0] = a.bias[t];
        if (a.scale) {
            pv[1] = a.scale[t];
            pv[2] = a.shift[t];
        }
    }

    int tb, ty0, tx0;  // image and first output pixel of a tile
    auto locate = [&](int tile) {
        tb = (int)fdiv((unsigned)tile, a.div_tpi);
        const int rem = tile - tb * (int)a.div_tpi.d;
        const int tx = (int)fdiv((unsigned)rem, a.div_ty);
        ty0 = (rem - tx * a.tiles_y) * TH;
        tx0 = tx * TW;
    };

    const int nt = wave >> 2, mt = wave & 3;  // output rows 2 nt, 2 nt + 1 of the tile; output channels [32 mt, +32)
    const int r = 2 * nt + (lr >> 4), c = lr & 15;

    // ---------------------------------------------------------------------- the copies: wave w issues pieces w, w + 8, ...
    const i32x4 xr = make_buf(a.x, a.x_bytes);
    // per-lane constants: byte offset from input pixel (2 ty0 - 1, 2 tx0 - 1), and flags: bit 0 first staged row (above the
    // image for the top tiles), bit 1 first odd column (left of the image for the leftmost tiles), bit 2 nothing to copy
    int rel[MYP], flag[MYP];
#pragma unroll
    for (int j = 0; j < MYP; ++j) {
        const int p = (wave + NWC * j) * 8 + (lane >> 3);  // staged pixel: plane after plane
        const int pl = p >= IR * PC ? 1 : 0, n = p - pl * IR * PC;
        const int row = n / PC, jc = n - row * PC;
        const int s = (lane & 7) ^ ((jc >> 1) & 7);
        const int col = 2 * jc + pl;  // input column relative to 2 tx0 - 1
        rel[j] = (row * a.W + col) * (a.x_ld * 2) + s * 16;
        flag[j] = (row == 0 ? 1 : 0) | (col == 0 ? 2 : 0) | (row >= IR || col == IC - 1 ? 4 : 0);
    }
    auto issue = [&](int tile, int buf) {
        locate(tile);
        const int base = ((tb * a.H + 2 * ty0 - 1) * a.W + 2 * tx0 - 1) * (a.x_ld * 2);
        const int dead = 4 | (ty0 == 0 ? 1 : 0) | (tx0 == 0 ? 2 : 0);
#pragma unroll
        for (int j = 0; j < MYP; ++j)
            dma16_buf(xr, (flag[j] & dead) ? Y5_BUF_OOB : (unsigned)(base + rel[j]), (unsigned)(buf * X_BYTES + (wave + NWC * j) * 1024));
    };

    int uoff[9];  // byte offset of this lane's K-step-0 fragment per tap; K-step ks: offset ^ (ks << 5)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int row = 2 * r + tap / 3, dx = tap % 3;
        uoff[tap] = dx == 1 ? swz(1, row, c, lh) : swz(0, row, c + (dx == 2 ? 1 : 0), lh);
    }
    float negzero;
    asm volatile("v_mov_b32 %0, 0x80000000" : "=v"(negzero));

    // The first tile's pieces go out before the weights are fetched, so that both arrive in one round trip (the weights
    // first, then the copies, cost the launch two).
    if (first < last) issue(first, 0);
    f16x8 wf[9 * CI / 16];
#pragma unroll
    for (int ks = 0; ks < 9 * CI / 16; ++ks) wf[ks] = *reinterpret_cast<const f16x8*>(a.w + (size_t)(mt * 32 + perm_row(lr)) * a.w_ld + ks * 16 + lh * 8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // from here on the only loads in flight are the copies counted below
#pragma unroll
    for (int ks = 0; ks < 9 * CI / 16; ++ks) asm volatile("" : "+v"(wf[ks]));  // in registers: nothing in the tile loop waits for them
    if (t < CO) {
        prm[t] = pv[0];
        prm[CO + t] = pv[1];
        prm[2 * CO + t] = pv[2];
    }
    // two tiles ahead: when tile i's pieces are waited for, those of tile i + 1 (the newest MYP) may still be in flight
    if (first + 1 < last) issue(first + 1, 1);
    int buf = 0;
    for (int tile = first; tile < last; ++tile) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave's pieces of this tile have landed; tile - 1 is read
        if (tile + 2 < last) issue(tile + 2, buf == 0 ? NBUF - 1 : buf - 1);  // the buffer that held tile - 1
        locate(tile);
        const unsigned char* X = smem + buf * X_BYTES;
        f32x16 acc = zero16();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < CI / 16; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[tap * (CI / 16) + ks], *reinterpret_cast<const f16x8*>(X + (uoff[tap] ^ (ks << 5))), acc, 0, 0, 0);
        f16x8 lo, hi;
        finish<F>(acc, prm, CO, mt * 32 + 16 * lh, a.slope, negzero, lo, hi);
        f16* yp = a.y + ((size_t)(tb * a.Ho + ty0 + r) * a.Wo + tx0 + c) * a.y_ld + mt * 32 + 16 * lh;
        // the next tile's pieces (issued a whole tile ago) must have landed before the next barrier; the pieces issued at the
        // top of this tile are younger and may stay in flight, this tile's stores are issued after the wait
        if (tile + 2 < last)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MYP) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

template <int F>
int launch(const S2Args& a, hipStream_t s) {
    static std::atomic<unsigned long long> attr_set{0};
    const unsigned long long bit = 1ull << y5_current_device();
    if (!(attr_set.load() & bit)) {
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_s2c64_kernel<F>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        attr_set.fetch_or(bit);
    }
    const int cus = cu_count();
    hipLaunchKernelGGL((conv_s2c64_kernel<F>), dim3((unsigned)(a.ntiles < cus ? a.ntiles : cus)), dim3(NTH), (size_t)SMEM, s, a);
    Y5_LAUNCH_CHECK("conv_s2c64_kernel");
    return Y5_OK;
}

}  // namespace

int y5_conv_s2c64_count() { return 1; }

// Returns 0 when the convolution is outside what this kernel is compiled for (then *rc is untouched), 1 after a launch
// attempt with its status in *rc.  d was validated by y5_launch_conv.
int y5_try_conv_s2c64(const y5_conv_desc& d, hipStream_t s, int* rc) {
    if (d.ksize != 3 || d.stride != 2 || d.pad != 1 || d.cin != CI || d.cout != CO || d.res) return 0;
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
    *rc = f == 0 ? launch<0>(a, s) : f == 1 ? launch<1>(a, s) : f == 2 ? launch<2>(a, s) : launch<3>(a, s);
    return 1;
}
