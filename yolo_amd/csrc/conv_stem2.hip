// Focus + its 3x3 convolution (3 -> 32 channels at half resolution) + the following 3x3 stride-2 convolution 32 -> 64, as ONE
// persistent kernel: yolov5s layers 0 and 1 (models/common.py:79-86, :26; models/yolov5s.yaml backbone).  The 32-channel
// tensor between the two layers is the largest activation of the graph (6.5 MB per 640x640 image, written once and read
// once); here it never leaves the chip.
//
// Per 8 x 16 tile of layer 1's output: the eight waves share the copy of the 38 x 80 x 3 image patch global -> LDS (one
// tile ahead, two or three 1 KiB pieces each), then multiply the Focus convolution for the 17 x 33 pixels of layer 0's output that the tile's
// stride-2 windows touch - as the 6x6 / stride-2 convolution over the image that Focus + conv amount to,
// K ordered (image row, channel, 8 columns) exactly as in conv_focus.hip, so the sums are bit-identical to that kernel's -
// apply layer 0's post-ops and write the result into two LDS planes (odd / even columns, zeros where the window leaves
// layer 0's output: layer 1's padding); then they multiply layer 1's nine taps from those planes exactly as conv_s2c32.hip
// does and store from registers.  Both layers' weights live in registers (eight waves of 64 lanes: 256 registers each),
// and a wave reads the image words of its next 32-pixel group while the current group multiplies.  Barriers are LDS-only;
// the one vmcnt wait per tile sits after layer 1's multiplies, a whole tile after the copy it waits for was issued.
#include <atomic>
#include <type_traits>

#include "y5_common.h"

namespace {

#include "conv_common.h"
#include "epi_mix.h"

struct StemArgs {
    const f16* img;   // [B][3][H][W]
    f16* y;           // layer 1's output (B, H/4, W/4, 64), pixel pitch y_ld
    const f16 *w0, *w1;
    const float *b0, *s0, *h0, *b1, *s1, *h1;
    int w0_ld, w1_ld, H, W, Ho, Wo, y_ld, img_bytes;
    int tiles_y, ntiles;
    FastDiv div_tpi, div_ty;
    float slope0, slope1;
};

constexpr int C0 = 32, C1 = 64;
constexpr int TH = 8, TW = 16;                                   // tile of layer 1's output
constexpr int FR = 2 * TH + 1, FC = 2 * TW + 1, FPX = FR * FC;   // layer 0 pixels a tile needs: 17 x 33
constexpr int PC = (FC + 1) / 2 + 1;                             // pixels per plane row: 17 columns + 1 (see plane_off)
constexpr int NT0 = (FPX + 31) / 32;                             // 32-pixel groups of them (18)
// image patch per channel: 38 rows of 40 pixel pairs = ten 16-byte slots, from image column 4 tx0 - 8 on (16-byte aligned;
// the windows use columns 4 tx0 - 4 .. 4 tx0 + 65, i.e. pairs 2 .. 36): every slot lies entirely inside or outside the image
constexpr int PR = 2 * FR + 4, PDW = 40, PSLOTS = PDW / 4, PCOL0 = 2;
constexpr int PATCH_PIECES = (3 * PR * PSLOTS + 63) / 64, PATCH_BYTES = PATCH_PIECES * 1024;
constexpr int W0ROW = 152;                                       // halfs per layer-0 weight row in LDS: 18 k-groups of 8 + 8 pad (19 slots: conflict-free)
constexpr int ROW = C0 * 2;                                      // bytes per layer-0 pixel in the planes
constexpr int PLANES_USED = 2 * FR * PC * ROW;                   // 39 168
constexpr int PLANES_BYTES = PLANES_USED + 64 * 32;              // + a 32-byte slot per lane for the lanes past the region
constexpr int NWC = 8, NTH = 64 * NWC;
constexpr int MYP = (PATCH_PIECES + NWC - 1) / NWC;              // patch pieces a wave copies (piece d belongs to wave d % 8)
constexpr int PATCH0 = 0, PLANES0 = 2 * PATCH_BYTES, PRM0 = PLANES0 + 2 * PLANES_BYTES, W00 = PRM0 + 3 * (C0 + C1) * 4;
constexpr int SMEM = W00 + C0 * W0ROW * 2;
static_assert(PLANES0 % 16 == 0 && PRM0 % 16 == 0 && W00 % 16 == 0 && SMEM <= 160 * 1024, "LDS layout");

// 16-byte slot s of layer-0 pixel (row, j) of plane pl (0: odd columns of the needed region, i.e. columns 2 tx0 - 1, + 1, ...;
// 1: even).  Layer 1 reads, per ds_read_b128 pass, eight pixels of one plane row and eight of the row two below, shifted by
// the tap: with a row pitch of 18 pixels (two rows = a multiple of four pixels = of 256 bytes) and the swizzle following the
// COLUMN, the 16 lanes land on 16 different bank groups for every tap (pitch 17 and a swizzle on the linear index: 2-way
// conflicts on every read)
__device__ __forceinline__ int plane_off(int pl, int row, int j, int s) { return ((pl * FR + row) * PC + j) * ROW + ((s ^ ((j >> 2) & 3)) << 4); }

#ifdef Y5_STAMPS
// Diagnostic build only: per compute wave, cycles in {barrier A, layer 0, barrier B, layer 1 + store} summed over tiles
__device__ unsigned long long y5_stem2_stamp_buf[256 * 8 * 8];
#define STEM_STAMP(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[i] += now_ - st_last; st_last = now_; }
#else
#define STEM_STAMP(i)
#endif

template <int F0, int F1, bool PIPE>
__global__ __launch_bounds__(NTH) void stem2_kernel(const StemArgs a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, lr = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nwg = gridDim.x, g = xcd_remap(blockIdx.x, nwg);
    const int first = (int)((long long)g * a.ntiles / nwg), last = (int)((long long)(g + 1) * a.ntiles / nwg);

    int tb, ty0, tx0;  // image and first layer-1 output pixel of a tile
    auto locate = [&](int tile) {
        tb = (int)fdiv((unsigned)tile, a.div_tpi);
        const int rem = tile - tb * (int)a.div_tpi.d;
        const int tx = (int)fdiv((unsigned)rem, a.div_ty);
        ty0 = (rem - tx * a.tiles_y) * TH;
        tx0 = tx * TW;
    };

    // ------------------------------------------------------------------ the image patch, [channel][row][pair]
    const i32x4 ir = make_buf(a.img, a.img_bytes);
    // piece d, lane: slot index d * 64 + lane -> (channel, patch row, slot of 8 columns); byte offset from image element
    // (channel 0, row 4 ty0 - 4, column 4 tx0 - 8).  Wave w copies pieces NWC - 1 - w, + NWC, ... (the waves with one more
    // 32-pixel group of layer 0 copy one piece fewer)
    int rel[MYP], rc[MYP];
#pragma unroll
    for (int j = 0; j < MYP; ++j) {
        const int d = NWC - 1 - wave + NWC * j;
        const int idx = d * 64 + lane, ch = idx / (PR * PSLOTS), rem = idx - ch * (PR * PSLOTS);
        const int row = rem / PSLOTS, sl = rem - row * PSLOTS;
        const bool unused = ch >= 3 || d >= PATCH_PIECES;  // slots past the third channel: read as out of range, i.e. zeros
        rel[j] = unused ? 0x7fffffff : ((ch * a.H + row) * a.W + 8 * sl) * 2;  // base + this stays below 2^32 and at or above img_bytes
        rc[j] = row | (sl << 8) | (unused ? 1 << 16 : 0);
    }
    auto issue = [&](int tile, int buf) {
        locate(tile);
        const int iy0 = 4 * ty0 - 4, ix0 = 4 * tx0 - 8;
        const int base = ((tb * 3 * a.H + iy0) * a.W + ix0) * 2;
        const bool inner = iy0 >= 0 && iy0 + PR <= a.H && ix0 >= 0 && ix0 + 2 * PDW <= a.W;
        if (inner) {  // no clipping: a lane's address is the patch origin plus its fixed offset
#pragma unroll
            for (int j = 0; j < MYP; ++j) {
                const int d = NWC - 1 - wave + NWC * j;
                if (d >= PATCH_PIECES) break;
                dma16_buf(ir, (unsigned)(base + rel[j]), (unsigned)(PATCH0 + buf * PATCH_BYTES + d * 1024));
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < MYP; ++j) {
            const int d = NWC - 1 - wave + NWC * j;
            if (d >= PATCH_PIECES) break;
            const int row = rc[j] & 255, sl = (rc[j] >> 8) & 255;
            const bool ok = !(rc[j] >> 16) && (unsigned)(iy0 + row) < (unsigned)a.H && (unsigned)(ix0 + 8 * sl) < (unsigned)a.W;
            dma16_buf(ir, ok ? (unsigned)(base + rel[j]) : Y5_BUF_OOB, (unsigned)(PATCH0 + buf * PATCH_BYTES + d * 1024));
        }
    };
    if (first < last) issue(first, 0);

    // ------------------------------------------------------------------ parameters and layer 0's weights -> LDS
    // All of a thread's loads are issued (after the first patch's copies) before the first is waited for: written as
    // load -> store loops, the staging was eleven dependent round trips to L2 / the Infinity Cache at the start of the launch.
    float* prm = reinterpret_cast<float*>(smem + PRM0);  // [bias | scale | shift] of layer 0 (C0 each), then of layer 1 (C1 each)
    f16* w0l = reinterpret_cast<f16*>(smem + W00);
    {
        static_assert(C0 <= NTH && C1 <= NTH && (C0 * 144) % NTH == 0, "one parameter per thread, whole rounds of weights");
        float p0[3] = {0.f, 1.f, 0.f}, p1[3] = {0.f, 1.f, 0.f};
        if (t < C0) {
            if (a.b0) p0[0] = a.b0[t];
            if (a.s0) { p0[1] = a.s0[t]; p0[2] = a.h0[t]; }
        }
        if (t < C1) {
            if (a.b1) p1[0] = a.b1[t];
            if (a.s1) { p1[1] = a.s1[t]; p1[2] = a.h1[t]; }
        }
        // layer 0 weights: row i = output channel perm(i); k-group g = r6 * 3 + c (image row r6 of the 6x6 window, channel c)
        // holds the 8 columns q6 (6 used) of the packed row k = (r*3 + q)*16 + (dy + 2*dx)*3 + c with r6 = 2r + dy, q6 = 2q + dx
        constexpr int ROUNDS = C0 * 144 / NTH;
        f16 wv[ROUNDS];
#pragma unroll
        for (int j = 0; j < ROUNDS; ++j) {
            const int idx = t + j * NTH, n = idx / 144, rem = idx - n * 144;
            const int grp = rem >> 3, q6 = rem & 7, r6 = grp / 3, c = grp - 3 * r6;
            wv[j] = q6 < 6 ? a.w0[(size_t)perm_row(n) * a.w0_ld + ((r6 >> 1) * 3 + (q6 >> 1)) * 16 + ((r6 & 1) + 2 * (q6 & 1)) * 3 + c] : (f16)0.f;
        }
        if (t < C0) { prm[t] = p0[0]; prm[C0 + t] = p0[1]; prm[2 * C0 + t] = p0[2]; }
        if (t < C1) { prm[3 * C0 + t] = p1[0]; prm[3 * C0 + C1 + t] = p1[1]; prm[3 * C0 + 2 * C1 + t] = p1[2]; }
#pragma unroll
        for (int j = 0; j < ROUNDS; ++j) {
            const int idx = t + j * NTH, n = idx / 144, rem = idx - n * 144;
            w0l[n * W0ROW + rem] = wv[j];
        }
    }

    // ---------------------------------------------------------------------- weights and per-lane offsets
    __syncthreads();
    // MFMA step s of layer 0 multiplies k-groups s (lanes 0-31) and s + 9 (lanes 32-63)
    f16x8 w0f[9];
#pragma unroll
    for (int s9 = 0; s9 < 9; ++s9) w0f[s9] = *reinterpret_cast<const f16x8*>(w0l + lr * W0ROW + (9 * lh + s9) * 8);
    const int nt1 = wave >> 1, mt1 = wave & 1;  // layer 1: output rows 2 nt1, 2 nt1 + 1 of the tile; channels [32 mt1, +32)
    const int r1 = 2 * nt1 + (lr >> 4), c1 = lr & 15;
    f16x8 w1f[9 * C0 / 16];
#pragma unroll
    for (int ks = 0; ks < 9 * C0 / 16; ++ks) w1f[ks] = *reinterpret_cast<const f16x8*>(a.w1 + (size_t)(mt1 * 32 + perm_row(lr)) * a.w1_ld + ks * 16 + lh * 8);
    int uoff[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int row = 2 * r1 + tap / 3, dx = tap % 3;
        uoff[tap] = dx == 1 ? plane_off(1, row, c1, lh) : plane_off(0, row, c1 + (dx == 2 ? 1 : 0), lh);
    }
    // layer 0: this wave's 32-pixel groups wave, wave + 8, (wave + 16) of the 17 x 33 region; pixel n = 32 group + lr sits
    // in row frow, column fcol (row fy = 2 ty0 - 1 + frow, column fx = 2 tx0 - 1 + fcol of layer 0's output)
    constexpr int NU = (NT0 + NWC - 1) / NWC;
    int boff[NU], poff[NU], edge = 0;  // edge: bit 2 i = group i's pixel is in the region's first row, bit 2 i + 1 = in its first column
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int n = (wave + NWC * i) * 32 + lr;
        const int frow = min(n / FC, FR - 1), fcol = n - (n / FC) * FC;  // lanes past the region repeat its last row (never written)
        // patch rows 2 frow + r6, pair columns fcol .. fcol + 3 (8 image columns from 2 fcol on)
        boff[i] = ((2 * frow + 3 * lh) * PDW + fcol + PCOL0) * 4;
        poff[i] = n < FPX ? plane_off(fcol & 1, frow, fcol >> 1, 2 * lh) : PLANES_USED + lane * 32;  // past the region: a slot nobody reads
        edge |= ((frow == 0 ? 1 : 0) | (fcol == 0 ? 2 : 0)) << (2 * i);
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    struct __attribute__((packed, aligned(4))) Quad { unsigned q0, q1, q2, q3; };
    float negzero;
    asm volatile("v_mov_b32 %0, 0x80000000" : "=v"(negzero));
    auto layer0 = [&](auto nu_c, const unsigned char* patch, unsigned char* planes, int zmask) {
        constexpr int N = decltype(nu_c)::value;
        // the image words of multiply g = 9 group + s9 (channel s9 % 3, window rows s9 / 3 in lanes 0-31 and 3 + s9 / 3 in lanes
        // 32-63) sit in slot g % RING; a slot is refilled for multiply g + RING right after multiply g has read it
        constexpr int RING = 3;
        u32x4 pd[RING];
        auto fetch = [&](int g) {
            const int i = g / 9, s9 = g % 9;
            const Quad q = *reinterpret_cast<const Quad*>(patch + boff[i] + ((s9 % 3) * PR + s9 / 3) * PDW * 4);
            pd[g % RING] = u32x4{q.q0, q.q1, q.q2, q.q3};
        };
        auto put = [&](int i, const f32x16& acc) {
            f16x8 lo, hi;
            finish<F0>(acc, prm, C0, 16 * lh, a.slope0, negzero, lo, hi);
            const unsigned keep = ((edge >> (2 * i)) & zmask) ? 0u : ~0u;
            *reinterpret_cast<u32x4*>(planes + poff[i]) = __builtin_bit_cast(u32x4, lo) & keep;
            *reinterpret_cast<u32x4*>(planes + (poff[i] ^ 16)) = __builtin_bit_cast(u32x4, hi) & keep;
        };
#pragma unroll
        for (int g = 0; g < RING; ++g) fetch(g);
        f32x16 acc[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            acc[i] = zero16();
#pragma unroll
            for (int s9 = 0; s9 < 9; ++s9) {
                const int g = 9 * i + s9;
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0f[s9], __builtin_bit_cast(f16x8, pd[g % RING]), acc[i], 0, 0, 0);
                if (g + RING < 9 * N) fetch(g + RING);
            }
            if (i > 0) put(i - 1, acc[i - 1]);
#pragma unroll
            for (int s9 = 0; s9 < 9; ++s9) {  // spread the previous group's post-ops and the operand refills over the multiplies
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (9 * i + s9 + RING < 9 * N) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                if (i > 0) __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        put(N - 1, acc[N - 1]);
    };
    int cur = 0;
#ifdef Y5_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#endif
    auto do_l0 = [&](int tile, int buf) {  // layer 0 of a tile: patch[buf] -> planes[buf]
        locate(tile);
        const unsigned char* patch = smem + PATCH0 + buf * PATCH_BYTES;
        unsigned char* planes = smem + PLANES0 + buf * PLANES_BYTES;
        const int zmask = (ty0 == 0 ? 1 : 0) | (tx0 == 0 ? 2 : 0);  // rows / columns outside layer 0's output: layer 1's zero padding
        // straight-line code per wave class, so that a group's multiplies overlap the previous group's post-ops
        const bool full = wave + NWC * (NU - 1) < NT0;
        if (full) layer0(std::integral_constant<int, NU>{}, patch, planes, zmask);
        else layer0(std::integral_constant<int, NU - 1>{}, patch, planes, zmask);
    };
    auto do_l1 = [&](int tile, int buf, bool wait_copies) {  // layer 1 of a tile from planes[buf], and its stores
        locate(tile);
        const unsigned char* planes = smem + PLANES0 + buf * PLANES_BYTES;
        f32x16 acc = zero16();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < C0 / 16; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1f[tap * (C0 / 16) + ks], *reinterpret_cast<const f16x8*>(planes + (uoff[tap] ^ (ks << 5))), acc, 0, 0, 0);
        f16x8 lo, hi;
        finish<F1>(acc, prm + 3 * C0, C1, mt1 * 32 + 16 * lh, a.slope1, negzero, lo, hi);
        f16* yp = a.y + ((size_t)(tb * a.Ho + ty0 + r1) * a.Wo + tx0 + c1) * a.y_ld + mt1 * 32 + 16 * lh;
        if (wait_copies) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next patch's pieces, issued most of a tile ago; own stores are never waited for here
        *reinterpret_cast<f16x8*>(yp) = lo;
        *reinterpret_cast<f16x8*>(yp + 8) = hi;
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the first patch
    if constexpr (PIPE) {
        // Software pipeline over the tiles with ONE barrier per tile: between two barriers every wave does layer 0 of tile i
        // (patch[cur] -> planes[cur]) and layer 1 of tile i - 1 (planes[cur ^ 1]) - waves 0-3 in that order, waves 4-7 in the
        // other.  Waves w and w + 4 share a SIMD: one multiplies layer 1 while the other runs layer 0's mix of multiplies and
        // post-ops, where in tile-by-tile order both are always in the same phase and matrix pipe and VALU take turns.
        const bool l1_first = wave >= NWC / 2;
        for (int tile = first; tile <= last; ++tile, cur ^= 1) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // patch[cur] and planes[cur ^ 1] complete, patch[cur ^ 1] and planes[cur] free
            STEM_STAMP(0)
            if (tile + 1 < last) issue(tile + 1, cur ^ 1);
            STEM_STAMP(4)
            if (!l1_first) {
                if (tile < last) do_l0(tile, cur);
                STEM_STAMP(1)
                if (tile > first) do_l1(tile - 1, cur ^ 1, true);
                STEM_STAMP(3)
            } else {
                if (tile > first) do_l1(tile - 1, cur ^ 1, false);
                STEM_STAMP(3)
                if (tile < last) do_l0(tile, cur);
                STEM_STAMP(1)
            }
            // the pieces of the next patch must have landed before the next barrier: waves that ran layer 1 last waited before its stores
            if (l1_first || tile == first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else {
        for (int tile = first; tile < last; ++tile, cur ^= 1) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // A: the patch is complete, the other buffers are free
            STEM_STAMP(0)
            if (tile + 1 < last) issue(tile + 1, cur ^ 1);  // that buffer was last read by layer 0 of tile - 1
            STEM_STAMP(4)
            do_l0(tile, cur);
            STEM_STAMP(1)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // B: the planes are complete
            STEM_STAMP(2)
            do_l1(tile, cur, true);
            STEM_STAMP(3)
        }
    }
#ifdef Y5_STAMPS
    if (lane == 0 && blockIdx.x < 256) {
        unsigned long long* o = y5_stem2_stamp_buf + (blockIdx.x * 8 + wave) * 8;
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

template <int F0, int F1, bool PIPE>
int launch(const StemArgs& a, hipStream_t s) {
    static std::atomic<unsigned long long> attr_set{0};
    const unsigned long long bit = 1ull << y5_current_device();
    if (!(attr_set.load() & bit)) {
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(stem2_kernel<F0, F1, PIPE>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        attr_set.fetch_or(bit);
    }
    const int cus = cu_count();
    hipLaunchKernelGGL((stem2_kernel<F0, F1, PIPE>), dim3((unsigned)(a.ntiles < cus ? a.ntiles : cus)), dim3(NTH), (size_t)SMEM, s, a);
    Y5_LAUNCH_CHECK("stem2_kernel");
    return Y5_OK;
}

}  // namespace

#ifdef Y5_STAMPS
extern "C" int y5_debug_stamps_stem2(unsigned long long* dst, int n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(y5_stem2_stamp_buf), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif

// f: the Focus convolution as y5_focus_conv_f16 takes it (x = the fp16 NCHW image, y ignored); g: the 3x3 stride-2
// convolution of its output (x ignored).
int y5_launch_focus_conv2(const y5_conv_desc& f, const y5_conv_desc& g, hipStream_t s) {
    Y5_REQUIRE(f.dtype == Y5_F16 && g.dtype == Y5_F16, "focus_conv2: fp16 only");
    Y5_REQUIRE(f.x && f.w && g.w && g.y, "focus_conv2: null image, weight or output pointer");
    Y5_REQUIRE(f.cin == 3 && f.ksize == 3 && f.stride == 1 && f.pad == 1 && f.cout == C0, "focus_conv2: the first layer is Focus + 3x3 stride 1 with %d filters", C0);
    Y5_REQUIRE(g.ksize == 3 && g.stride == 2 && g.pad == 1 && g.cin == C0 && g.cout == C1, "focus_conv2: the second layer is 3x3 stride 2, %d -> %d channels", C0, C1);
    Y5_REQUIRE(f.batch > 0 && g.batch == f.batch && f.h % 4 == 0 && f.w_in % 4 == 0 && f.ho == f.h / 2 && f.wo == f.w_in / 2 && g.h == f.ho && g.w_in == f.wo &&
                   g.ho == f.h / 4 && g.wo == f.w_in / 4, "focus_conv2: inconsistent geometry");
    Y5_REQUIRE(g.ho % TH == 0 && g.wo % TW == 0, "focus_conv2: output height must be a multiple of %d and width of %d (got %d x %d)", TH, TW, g.ho, g.wo);
    Y5_REQUIRE(f.kpad >= 144 && f.w_rows >= C0 && g.kpad >= 9 * C0 && g.w_rows >= C1, "focus_conv2: weights not packed (Focus: 16-channel space-to-depth layout)");
    Y5_REQUIRE(!f.res && !g.res, "focus_conv2: no residual input");
    Y5_REQUIRE(g.y_ld >= C1 && g.y_ld % 8 == 0 && y5_aligned16(g.y) && y5_aligned16(g.w) && (reinterpret_cast<uintptr_t>(f.x) & 3) == 0 && (reinterpret_cast<uintptr_t>(f.w) & 1) == 0,
               "focus_conv2: misaligned pointer or pitch");
    for (const y5_conv_desc* d : {&f, &g}) {
        Y5_REQUIRE((d->scale == nullptr) == (d->shift == nullptr), "focus_conv2: scale and shift come together");
        Y5_REQUIRE(d->act == Y5_ACT_NONE || (d->act == Y5_ACT_LEAKY && d->slope > 0.f && d->slope <= 1.f), "focus_conv2: bad activation");
    }
    const long long ib = (long long)f.batch * 3 * f.h * f.w_in * 2;
    Y5_REQUIRE(ib < 0x7fffffffLL, "focus_conv2: image batch of 2 GiB or more");
    StemArgs a{};
    a.img = static_cast<const f16*>(f.x);
    a.y = static_cast<f16*>(g.y);
    a.w0 = static_cast<const f16*>(f.w);
    a.w1 = static_cast<const f16*>(g.w);
    a.b0 = f.bias; a.s0 = f.scale; a.h0 = f.shift; a.b1 = g.bias; a.s1 = g.scale; a.h1 = g.shift;
    a.w0_ld = f.kpad; a.w1_ld = g.kpad; a.H = f.h; a.W = f.w_in; a.Ho = g.ho; a.Wo = g.wo; a.y_ld = g.y_ld; a.img_bytes = (int)ib;
    a.tiles_y = g.ho / TH;
    const int tiles_x = g.wo / TW;
    const long long total = (long long)f.batch * a.tiles_y * tiles_x;
    Y5_REQUIRE(total > 0 && total < 0x7fffffffLL, "focus_conv2: too many tiles");
    a.ntiles = (int)total;
    a.div_tpi = make_fastdiv((unsigned)(a.tiles_y * tiles_x));
    a.div_ty = make_fastdiv((unsigned)a.tiles_y);
    a.slope0 = f.act == Y5_ACT_LEAKY ? f.slope : 1.f;
    a.slope1 = g.act == Y5_ACT_LEAKY ? g.slope : 1.f;
    const int f0 = (f.bias ? 1 : 0) | (f.scale ? 2 : 0), f1 = (g.bias ? 1 : 0) | (g.scale ? 2 : 0);
    Y5_REQUIRE(f.variant == 0 || f.variant == 1, "focus_conv2: variant %d (0: tile by tile, 1: layer 1 of a tile beside layer 0 of the next)", f.variant);
    if (f0 == 1 && f1 == 1) return f.variant ? launch<1, 1, true>(a, s) : launch<1, 1, false>(a, s);
    if (f0 == 2 && f1 == 2) return f.variant ? launch<2, 2, true>(a, s) : launch<2, 2, false>(a, s);
    return y5_fail(Y5_EINVAL, "focus_conv2: compiled for bias-only layers (fused model) or scale+shift-only layers (BatchNorm), got %d / %d", f0, f1);
}
