// Weight-stationary 1x1 stride-1 convolution for gfx950 (K <= 256): a persistent workgroup keeps its BN x K weight
// slab in LDS for its whole life and streams BM-pixel activation tiles through a ring of LDS stages filled by
// global->LDS DMA NST-1 tiles ahead.  A tile needs no K loop over memory: all of K is resident, so the only
// barrier is the one that publishes a landed tile; the epilogue is private to each wave (its own LDS scratch),
// so waves drift apart and overlap each other's DMA waits.  Used for the many memory-bound 1x1 layers, where the
// im2col kernel pays a fresh prologue, weight fetch and exposed DMA latency per 64..128-pixel workgroup.
//
// LDS rows hold all K of one pixel / one output channel (KP*2 bytes).  16-byte chunk c of row r sits in slot
// c ^ f(r) with f = r%16 (KP >= 128), (r/2)%8 (KP = 64) or (r/4)%4 (KP = 32): the 16 rows a ds_read_b128 lane
// group touches fall in 16 distinct slots of the 256-byte bank row.
#include "y5_common.h"

namespace {

#include "conv_common.h"

template <int KP>
__device__ __forceinline__ int row_swz(int row) {
    return KP >= 128 ? (row & 15) : KP == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3);
}

// Wait until at most n vector-memory instructions of this wave (DMA and stores, which retire in issue order) are
// outstanding.  No barrier.
__device__ __forceinline__ void wait_vm(int n) {
#define Y5_WAIT_CASE(N) \
    case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (n) {
        Y5_WAIT_CASE(1) Y5_WAIT_CASE(2) Y5_WAIT_CASE(3) Y5_WAIT_CASE(4) Y5_WAIT_CASE(5) Y5_WAIT_CASE(6) Y5_WAIT_CASE(7)
        Y5_WAIT_CASE(8) Y5_WAIT_CASE(9) Y5_WAIT_CASE(10) Y5_WAIT_CASE(11) Y5_WAIT_CASE(12) Y5_WAIT_CASE(13)
        Y5_WAIT_CASE(14) Y5_WAIT_CASE(15) Y5_WAIT_CASE(16) Y5_WAIT_CASE(17) Y5_WAIT_CASE(18) Y5_WAIT_CASE(19)
        Y5_WAIT_CASE(20) Y5_WAIT_CASE(21) Y5_WAIT_CASE(22) Y5_WAIT_CASE(23) Y5_WAIT_CASE(24)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef Y5_WAIT_CASE
}

template <int BM, int BN, int WM, int WN, int KP, int NST>
__global__ __launch_bounds__(256) void conv_ws1x1_kernel(const ConvArgs a) {
    static_assert(WM * WN == 4, "four waves per workgroup");
    constexpr int WTM = BM / WM, WTN = BN / WN;
    static_assert(WTM == 32, "one 32-pixel MFMA tile per wave");
    constexpr int RN = WTN / 32;
    constexpr int ROWB = KP * 2;                 // bytes per LDS row
    constexpr int CPR = KP / 8;                  // 16-byte chunks per row
    constexpr int RPI = 256 / CPR;               // rows one workgroup-wide DMA instruction covers
    constexpr int A_CH = BM / RPI;               // DMA instructions per wave and tile
    constexpr int W_CH = BN / RPI;               // ... for the weight slab
    static_assert(BM % RPI == 0 && BN % RPI == 0, "tiles are staged RPI rows at a time");
    constexpr int W_BYTES = BN * ROWB;
    constexpr int A_BYTES = BM * ROWB;
    constexpr int EROW = WTN * 2 + 16;           // per-wave epilogue scratch row pitch
    constexpr int E_BYTES = 32 * EROW;
    constexpr int A0 = W_BYTES, E0 = A0 + NST * A_BYTES, P0 = E0 + 4 * E_BYTES;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int nt = blockIdx.x % a.tiles_n;
    const int slot0 = blockIdx.x / a.tiles_n, nslots = gridDim.x / a.tiles_n;
    const int n0 = nt * BN;
    const int tiles_m = (int)((a.M + BM - 1) / BM);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

    // ---- per-channel epilogue parameters go to LDS first: an ordinary global load inside the tile loop would make
    // the compiler wait for every DMA in flight before it.
    float* prm = reinterpret_cast<float*>(smem + P0);
    if (t < BN) {
        const int n = n0 + t;
        const bool in = n < a.Cout;
        prm[t] = a.bias && in ? a.bias[n] : 0.f;
        prm[BN + t] = a.scale && in ? a.scale[n] : 1.f;
        prm[2 * BN + t] = a.scale && in ? a.shift[n] : 0.f;
    }
    const bool has_bias = a.bias != nullptr, has_scale = a.scale != nullptr;

    // ---- weight slab: once per workgroup
    const int row0 = t / CPR, slot = t % CPR;
#pragma unroll
    for (int i = 0; i < W_CH; ++i) {
        const int row = row0 + RPI * i;
        const int kc = slot ^ row_swz<KP>(row);
        const bool ok = kc * 8 < a.w_ld;  // weight rows are padded to 64, the slab to a power of two
        dma16(ok ? (const void*)(a.w + (size_t)(n0 + row) * a.w_ld + kc * 8) : (const void*)y5_zero_page, lds0 + i * 4096 + wave * 1024);
    }
    // ---- activation tile DMA state
    int akc[A_CH];
    const f16* aptr[A_CH];  // pixel (slot0*BM + row) of this thread's rows, advanced tile by tile
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const int row = row0 + RPI * i;
        akc[i] = slot ^ row_swz<KP>(row);
        aptr[i] = a.x + (size_t)(slot0 * BM + row) * a.x_ld + akc[i] * 8;
    }
    const size_t astep = (size_t)nslots * BM * a.x_ld;
    int issue_m = slot0 * BM;  // first pixel of the next tile to fetch
    auto issue_tile = [&](int stage) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const bool ok = issue_m + row0 + RPI * i < a.M && akc[i] * 8 < a.Cin;
            dma16(ok ? (const void*)aptr[i] : (const void*)y5_zero_page, lds0 + A0 + stage * A_BYTES + i * 4096 + wave * 1024);
            aptr[i] += astep;
        }
        issue_m += nslots * BM;
    };

    const int my_tiles = slot0 < tiles_m ? (tiles_m - slot0 + nslots - 1) / nslots : 0;
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < my_tiles) issue_tile(s);

    const int lr = lane & 31, lh = lane >> 5;
    const int arow = wm * 32 + lr;
    const int ua = lh ^ row_swz<KP>(arow);
    int ub[RN], brow[RN];
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        brow[j] = wn * WTN + j * 32 + lr;
        ub[j] = lh ^ row_swz<KP>(brow[j]);
    }
    unsigned char* escr = smem + E0 + wave * E_BYTES;
    constexpr int CPW = WTN / 8;  // 16-byte chunks per scratch row
    // Stores a full tile's epilogue issues per wave (a wave whose channels are all padding issues none).
    const int nst_full = n0 + wn * WTN < a.Cout ? 32 * CPW / 64 : 0;

    for (int ti = 0; ti < my_tiles; ++ti) {
        int ahead = my_tiles - 1 - ti;
        if (ahead > NST - 2) ahead = NST - 2;
        // Younger than the tile needed now: `ahead` tiles of DMA and the stores of up to NST-1 finished tiles (only the
        // last tile of a workgroup can be a partial one, and nothing waits after it).
        wait_vm(ahead * A_CH + (ti < NST - 1 ? ti : NST - 1) * nst_full);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (ti + NST - 1 < my_tiles) issue_tile((ti + NST - 1) % NST);

        const unsigned char* A = smem + A0 + (ti % NST) * A_BYTES + arow * ROWB;
        f32x16 acc[RN];
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KP / 16; ++ks) {
            const f16x8 xf = *reinterpret_cast<const f16x8*>(A + (((2 * ks) ^ ua) << 4));
#pragma unroll
            for (int j = 0; j < RN; ++j) {
                const f16x8 wf = *reinterpret_cast<const f16x8*>(smem + brow[j] * ROWB + (((2 * ks) ^ ub[j]) << 4));
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, xf, acc[j], 0, 0, 0);
            }
        }

        // ---- epilogue, private to the wave: post-ops -> own LDS scratch -> 16-byte stores
        const int m0 = (slot0 + ti * nslots) * BM + wm * 32;
#pragma unroll
        for (int j = 0; j < RN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = j * 32 + 8 * g + 4 * lh;
                const int pl = wn * WTN + nl;  // padding channels carry neutral parameters and stay in the scratch
                f32x4 v = {acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
                if (has_bias) v += *reinterpret_cast<const f32x4*>(prm + pl);
                f16x4 h = __builtin_convertvector(v, f16x4);
                if (has_scale)
                    h = __builtin_convertvector(__builtin_convertvector(h, f32x4) * *reinterpret_cast<const f32x4*>(prm + BN + pl) +
                                                    *reinterpret_cast<const f32x4*>(prm + 2 * BN + pl),
                                                f16x4);
                *reinterpret_cast<f16x4*>(escr + lr * EROW + nl * 2) = leaky4_f16(h, a.slope);
            }
        }
#pragma unroll
        for (int p = 0; p < 32 * CPW / 64; ++p) {
            const int e = p * 64 + lane;
            const int row = e / CPW, cc = e - row * CPW;
            const int m = m0 + row, n = n0 + wn * WTN + cc * 8;
            if (m < a.M && n < a.Cout) {
                *reinterpret_cast<f16x8*>(a.y + (size_t)m * a.y_ld + n) = *reinterpret_cast<const f16x8*>(escr + row * EROW + cc * 16);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the weight DMA of a workgroup without tiles must land before exit
}

template <int BM, int BN, int WM, int WN, int KP, int NST>
int launch_ws(ConvArgs a, hipStream_t s) {
    constexpr int WTN = BN / WN;
    constexpr size_t lds = (size_t)BN * KP * 2 + (size_t)NST * BM * KP * 2 + 4 * 32 * (WTN * 2 + 16) + 3 * BN * 4;
    static_assert(lds <= 160 * 1024, "configuration does not fit in LDS");
    static std::atomic<unsigned long long> attr_set{0};  // one bit per device: the attribute is per function and device
    const unsigned long long attr_set_bit = 1ull << y5_current_device();
    auto kern = conv_ws1x1_kernel<BM, BN, WM, WN, KP, NST>;
    if (!(attr_set.load() & attr_set_bit)) {
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set.fetch_or(attr_set_bit);
    }
    a.tiles_n = (a.Cout + BN - 1) / BN;
    const long long tiles_m = (a.M + BM - 1) / BM;
    const int per_cu = (int)((160 * 1024) / lds) > 4 ? 4 : (int)((160 * 1024) / lds);
    long long slots = (256LL * per_cu) / a.tiles_n;   // persistent workgroups per channel tile
    if (slots < 1) slots = 1;
    if (slots > tiles_m) slots = tiles_m;
    const long long grid = slots * a.tiles_n;
    a.tiles_total = (int)grid;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, a);
    Y5_LAUNCH_CHECK("conv_ws1x1_kernel");
    return Y5_OK;
}

}  // namespace

// which = 0: choose the channel width by rule; 1..3: BN = 128, 64, 32.  Returns 1 and sets *rc when this kernel took
// the convolution, 0 when it does not apply (not 1x1 stride 1, K > 256, or a residual input).
int y5_try_conv_ws1x1(const y5_conv_desc& d, hipStream_t s, int which, int* rc) {
    if (d.ksize != 1 || d.stride != 1 || d.cin > 256 || d.res) return 0;
    ConvArgs a{};
    a.x = static_cast<const f16*>(d.x);
    a.w = static_cast<const f16*>(d.w);
    a.bias = d.bias; a.scale = d.scale; a.shift = d.shift;
    a.res = static_cast<const f16*>(d.res);
    a.y = static_cast<f16*>(d.y);
    a.M = (long long)d.batch * d.ho * d.wo;
    a.H = d.h; a.W = d.w_in; a.Cin = d.cin; a.x_ld = d.x_ld;
    a.Ho = d.ho; a.Wo = d.wo; a.Cout = d.cout; a.y_ld = d.y_ld; a.res_ld = d.res_ld;
    a.stride = 1; a.pad = 0; a.K = d.cin; a.w_ld = d.kpad;
    a.slope = d.act == Y5_ACT_LEAKY ? d.slope : 1.f;
    auto waste = [&](int bn) { return ((d.cout + bn - 1) / bn) * bn - d.cout; };
    int bn = 128;
    if (waste(64) < waste(bn)) bn = 64;
    if (waste(32) < waste(bn) && d.cout <= 32) bn = 32;
    if (which) bn = 256 >> which;
    const int kp = d.cin <= 32 ? 32 : d.cin <= 64 ? 64 : d.cin <= 128 ? 128 : 256;
#define Y5_WS(BM, BN, WM, WN, KP, NST) *rc = launch_ws<BM, BN, WM, WN, KP, NST>(a, s); return 1
    if (bn == 128) {
        if (kp == 32) { Y5_WS(64, 128, 2, 2, 32, 4); }
        if (kp == 64) { Y5_WS(64, 128, 2, 2, 64, 4); }
        if (kp == 128) { Y5_WS(64, 128, 2, 2, 128, 3); }
        Y5_WS(64, 128, 2, 2, 256, 2);
    }
    if (bn == 64) {
        if (kp == 32) { Y5_WS(64, 64, 2, 2, 32, 4); }
        if (kp == 64) { Y5_WS(64, 64, 2, 2, 64, 4); }
        if (kp == 128) { Y5_WS(64, 64, 2, 2, 128, 3); }
        Y5_WS(64, 64, 2, 2, 256, 2);
    }
    if (kp <= 64) { Y5_WS(128, 32, 4, 1, 64, 4); }
    if (kp == 128) { Y5_WS(128, 32, 4, 1, 128, 3); }
    Y5_WS(128, 32, 4, 1, 256, 2);
#undef Y5_WS
}
