// Phased implicit-GEMM convolution ("ph"): the kernel for the compute-bound layers (3x3 / 1x1, any stride of {1, 2},
// any channel counts that are multiples of 8; what models/yolo.py:195 and structured pruning produce), fp16 or bf16.
//
// GEMM view as in conv_igemm.hip: D[n][m] = sum_k W[n][k] * X[m][k], m = output pixel, n = output channel,
// k = (tap, input channel) flattened - a 64-wide K-tile may span taps, so K = 9 * Cin is walked without per-tap padding
// (Cin = 80 costs 12 K-tiles, not 18).  What differs is the schedule:
//   * eight waves, each owning 128 pixels x 64 channels of the tile as 8 x 4 accumulators of v_mfma_f32_16x16x32
//     (WM x WN waves: 256 x 256 tiles for wide layers, 512 x 128 for 128-channel ones);
//   * a K-tile is staged as four half-tiles (pixel halves P0 / P1 = the rows of the waves' first / second 64 pixels,
//     weight halves W0 / W1 = the rows of their first / second 32 channels) into two rings of four slots, by
//     buffer_load ... lds issued by all waves, one half-tile per phase, up to 1.5 K-tiles ahead;
//   * a K-tile is multiplied in four phases, one 64 x 32 quadrant of the wave tile each (16 MFMAs): fragment reads of
//     the phase, one half-tile of DMAs, [barrier], 16 MFMAs, [barrier].  Waves 4-7 run one barrier behind waves 0-3, so
//     on every SIMD one wave multiplies while its partner reads LDS and issues DMAs;
//   * the only wait for DMAs is a counted s_waitcnt vmcnt(N) once per K-tile that leaves the three youngest half-tiles
//     in flight; barriers are LDS-only.
// Hazards.  RAW: a half-tile is read no earlier than the phase after the counted wait that retired it, behind a barrier
// every wave has passed after its own wait.  WAR: a slot is refilled one phase after its last read, and every phase
// drains its reads (lgkmcnt(0)) before its first barrier.  K-tiles past the end are "issued" with out-of-range offsets
// (the descriptor's range check turns them into zero fills of free slots), which keeps the wait count constant.
#include "y5_common.h"

namespace {

#include "conv_common.h"

template <typename ET>
struct Mma16;
template <>
struct Mma16<f16> {
    static __device__ __forceinline__ f32x4 run(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <>
struct Mma16<bf16> {
    static __device__ __forceinline__ f32x4 run(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

// WM x WN waves (eight), each owning MF x NF accumulators of 16 pixels x 16 channels: tile = (WM*MF*16) pixels x
// (WN*NF*16) channels.  MF and NF pick the tile that fills the chip in whole rounds for a layer's pixel and channel
// counts (the engine times the compiled shapes per layer); the halves of the schedule hold ceil / floor of them.
template <int WM, int WN, int MF, int NF>
struct PhCfg {
    static_assert(WM * WN == 8, "eight waves");
    static_assert(MF >= 2 && MF <= 8 && NF >= 2 && NF <= 6 && MF * NF <= 32, "at most 128 accumulator registers");
    static constexpr int MF0 = (MF + 1) / 2, MF1 = MF / 2, NF0 = (NF + 1) / 2, NF1 = NF / 2;
    static constexpr int BM = WM * MF * 16, BN = WN * NF * 16;
    static constexpr int PR = MF0 * 16, WR = NF0 * 16;            // rows a wave row / wave column owns in a slot
    static constexpr int P_SLOT = WM * PR * 128, W_SLOT = WN * WR * 128;  // bytes (128-byte rows)
    static constexpr int P_PIECES = WM * PR / 8, W_PIECES = WN * WR / 8;  // 1 KiB DMA pieces per half-tile
    static constexpr int NP = (P_PIECES + 7) / 8, NW = (W_PIECES + 7) / 8;  // DMA instructions per thread and half-tile
    static constexpr bool DUMP = (P_PIECES % 8) != 0 || (W_PIECES % 8) != 0;  // some waves issue a piece past the slot:
    static constexpr int RING = 4 * (P_SLOT + W_SLOT);                        // it zero-fills 1 KiB behind the rings
    static constexpr int EP = BN * 2 + 16;                       // staged output tile: row pitch in bytes
    static constexpr int TILE = BM * EP;
    static constexpr int NEED_LOOP = RING + (DUMP ? 1024 : 0), NEED_EPI = TILE + 3 * BN * 4;  // parameters behind the staged tile
    static constexpr int SMEM = NEED_LOOP > NEED_EPI ? NEED_LOOP : NEED_EPI;
    static_assert(SMEM <= 160 * 1024, "one workgroup per compute unit");
};

#define PH_SYNC_READS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define PH_BARRIER() asm volatile("s_barrier" ::: "memory")

template <typename ET, int WM, int WN, int MF, int NF>
__global__ __launch_bounds__(512, 2) void conv_ph_kernel(const ConvArgs a_unused) {
    typedef PhCfg<WM, WN, MF, NF> Cfg;
    typedef typename Elem<ET>::v8 ev8;
    typedef typename Elem<ET>::v4 ev4;
    constexpr int MF0 = Cfg::MF0, MF1 = Cfg::MF1, NF0 = Cfg::NF0, NF1 = Cfg::NF1;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, EP = Cfg::EP, PR = Cfg::PR, WR = Cfg::WR;
    constexpr int P_SLOT = Cfg::P_SLOT, W_SLOT = Cfg::W_SLOT, W_RING = 4 * P_SLOT;
    constexpr int NP = Cfg::NP, NW = Cfg::NW;
    constexpr int VM = 2 * NP + NW;           // three half-tiles (P, W, P) stay in flight at the wait of a K-tile
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int late = wave >> 2;  // waves 4-7: the SIMD partners of waves 0-3, one barrier behind them
    // The arguments are read from the kernel-argument segment where they are used (scalar loads behind an opaque copy of
    // its address) instead of living in scalar registers from the first instruction on: the K loop needs a dozen of
    // them, the per-tile set-up and the epilogue three dozen others, and all of them together do not fit.
    typedef const __attribute__((address_space(4))) ConvArgs* KArgs;
    KArgs ka = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();  // the kernel's only explicit argument starts the segment
    (void)a_unused;
#define PH_ARGS(NAME)                \
    asm volatile("" : "+s"(ka));     \
    const auto& NAME = *ka;
    PH_ARGS(a)
    const int nk = a.nk, tiles_total = a.tiles_total, tiles_n = a.tiles_n;
    const unsigned argK = (unsigned)a.K, argCin = (unsigned)a.Cin, cin_mul = a.div_cin.mul, cin_shr = a.div_cin.shr;
    const int argW = a.W, arg_xld = a.x_ld;

    // ---- DMA constants.  A 1 KiB piece covers 8 slot rows; lane l fills 16-byte slot (l & 7) of row (l >> 3), which
    // holds K-chunk (l & 7) ^ f(row), f(row) = (row >> 1) & 7.  A wave's pieces are wave, wave + 8, ...: their rows all
    // have the same f for a given lane, so one K-chunk index serves every DMA of the thread.
    const int kc = (lane & 7) ^ (((wave & 1) << 2) | (lane >> 4));
    const i32x4 xr = make_buf(a.x, a.x_bytes), wr = make_buf(a.w, a.w_bytes);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // ---- fragment addressing: lane l reads row (l & 15) of a 16-row group, K-chunk 4*ks + (l >> 4), at slot chunk ^ f(row)
    const int lr = lane & 15, lq = lane >> 4;
    const int sw = lr >> 1;  // f(row): the groups start at multiples of 16 rows
    const int coff0 = ((lq ^ sw) << 4), coff1 = (((4 + lq) ^ sw) << 4);
    const unsigned char* prow = smem + (wm * PR + lr) * 128;
    const unsigned char* wrow = smem + W_RING + (wn * WR + lr) * 128;
    float* prm = reinterpret_cast<float*>(smem + Cfg::TILE);

#define PH_READ_P(SLOT, N)                                                                           \
    _Pragma("unroll") for (int i = 0; i < (N); ++i) {                                                \
        pf[i][0] = *reinterpret_cast<const ev8*>(prow + (SLOT) * P_SLOT + i * 2048 + coff0);         \
        pf[i][1] = *reinterpret_cast<const ev8*>(prow + (SLOT) * P_SLOT + i * 2048 + coff1);         \
    }
#define PH_READ_W(WF, SLOT, N)                                                                       \
    _Pragma("unroll") for (int j = 0; j < (N); ++j) {                                                \
        WF[j][0] = *reinterpret_cast<const ev8*>(wrow + (SLOT) * W_SLOT + j * 2048 + coff0);         \
        WF[j][1] = *reinterpret_cast<const ev8*>(wrow + (SLOT) * W_SLOT + j * 2048 + coff1);         \
    }
#define PH_MMA(M0, MN, N0, NN, WF)                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    __builtin_amdgcn_s_setprio(1);                                                                   \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                 \
        _Pragma("unroll") for (int i = 0; i < (MN); ++i)                                             \
            _Pragma("unroll") for (int j = 0; j < (NN); ++j)                                         \
                acc[(M0) + i][(N0) + j] = Mma16<ET>::run(WF[j][ks], pf[i][ks], acc[(M0) + i][(N0) + j]); \
    __builtin_amdgcn_s_setprio(0);                                                                   \
    __builtin_amdgcn_sched_barrier(0);

    // persistent: a workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...
    for (int it = blockIdx.x; it < tiles_total; it += gridDim.x) {
        const int tile = xcd_remap(it, tiles_total);
        const int mt = tile / tiles_n, nt = tile - mt * tiles_n;
        const int m0 = mt * BM, n0 = nt * BN;
        EpiParams epi = {0.f, 1.f, 0.f};  // in registers through the K loop, staged in LDS behind it
        {
            PH_ARGS(a)
            const int n = n0 + t;
            if (t < BN && n < a.Cout) {
                if (a.bias) epi.b = a.bias[n];
                if (a.scale) {
                    epi.s = a.scale[n];
                    epi.h = a.shift[n];
                }
            }
        }

        // ---- DMA state of the tile
        int pbase[2][NP];       // byte offset of pixel (b, oy*s - pad, ox*s - pad) of the row (tap (0, 0); may be negative)
        unsigned pmask[NP];     // bit 16*h + tp: tap tp of the row of half h lies inside the image (1x1: tap 0 = row < M)
        unsigned wbase[2][NW];
        {
        PH_ARGS(a)
        const FastDiv div_hw = {a.div_hw.mul, a.div_hw.shr, a.div_hw.d}, div_wo = {a.div_wo.mul, a.div_wo.shr, a.div_wo.d};
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            pmask[i] = 0;
            const int p = wave + 8 * i;
            const bool pin = p < Cfg::P_PIECES;  // wave-uniform; false: the piece zero-fills the dump area
            const int r = p * 8 + (lane >> 3);   // row of the slot
            const int wrow_ = r / PR, within = r - wrow_ * PR;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int m = m0 + wrow_ * (MF * 16) + h * (MF0 * 16) + within;  // its pixel
                const bool mvalid = pin && within < (h ? MF1 : MF0) * 16 && m < a.M;
                const unsigned mm = mvalid ? (unsigned)m : 0u;
                const unsigned b = fdiv(mm, div_hw);
                const unsigned rem = mm - b * (unsigned)(a.Ho * a.Wo);
                const unsigned oy = fdiv(rem, div_wo), ox = rem - oy * (unsigned)a.Wo;
                const int hi0 = (int)oy * a.stride - a.pad, wi0 = (int)ox * a.stride - a.pad;
                pbase[h][i] = (((int)b * a.H + hi0) * a.W + wi0) * a.x_ld * 2;
                // taps inside the image: rows hi0 .. hi0+2 and columns wi0 .. wi0+2 (1x1: the pixel itself)
                unsigned rows = 0, cols = 0;
                const int span = a.pad ? 3 : 1;
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    if (d < span && (unsigned)(hi0 + d) < (unsigned)a.H) rows |= 1u << (3 * d);
                    if (d < span && (unsigned)(wi0 + d) < (unsigned)a.W) cols |= 1u << d;
                }
                pmask[i] |= (mvalid ? rows * cols : 0u) << (16 * h);  // bit 3r + q = rows bit 3r AND cols bit q
            }
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int p = wave + 8 * i;
            const bool pin = p < Cfg::W_PIECES;
            const int r = p * 8 + (lane >> 3);
            const int wcol = r / WR, within = r - wcol * WR;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int n = n0 + wcol * (NF * 16) + h * (NF0 * 16) + within;
                const bool nvalid = pin && within < (h ? NF1 : NF0) * 16;
                // rows past the packed weights fail the range check like the marker does: zeros
                wbase[h][i] = nvalid ? ((unsigned)n * (unsigned)a.w_ld + kc * 8) * 2u : Y5_BUF_OOB;
            }
        }
        }
        // position of this thread's chunk in K-tile kt: 1 << tap (0 when the chunk lies past K or the K-tile past the
        // last one) and the byte offset of (tap, channel) from tap (0, 0)
        unsigned pbit = 0;
        int poff = 0;
        auto decode = [&](int kt) {
            const unsigned k = (unsigned)kt * 64u + (unsigned)kc * 8u;
            const unsigned tp = __umulhi(k, cin_mul) >> cin_shr;  // k / Cin (Cin >= 8: never the divisor-1 case)
            const int c = (int)(k - tp * argCin);
            const int r = (int)(tp * 11u) >> 5, q = (int)tp - 3 * r;
            poff = ((r * argW + q) * arg_xld + c) * 2;
            pbit = (kt < nk && k < argK) ? 1u << (tp & 15u) : 0u;  // k < K: tap < 9
        };
        auto issue_p = [&](int h, int slot) {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const bool out = 8 * i + 7 >= Cfg::P_PIECES && wave + 8 * i >= Cfg::P_PIECES;  // only a slot's last pieces can lie past it
                const unsigned dst = out ? (unsigned)Cfg::RING : (unsigned)(slot * P_SLOT + (wave + 8 * i) * 1024);
                dma16_buf(xr, (pmask[i] & (pbit << (16 * h))) ? (unsigned)(pbase[h][i] + poff) : Y5_BUF_OOB, lds0 + dst);
            }
        };
        auto issue_w = [&](int h, int slot, int kt) {
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const bool out = 8 * i + 7 >= Cfg::W_PIECES && wave + 8 * i >= Cfg::W_PIECES;
                const unsigned dst = out ? (unsigned)Cfg::RING : (unsigned)(W_RING + slot * W_SLOT + (wave + 8 * i) * 1024);
                dma16_buf(wr, kt < nk ? wbase[h][i] + (unsigned)kt * 128u : Y5_BUF_OOB, lds0 + dst);
            }
        };

        ev8 pf[MF0][2], w0[NF0][2], w1[NF1 ? NF1 : 1][2];
        f32x4 acc[MF][NF];
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // ---- prologue: K-tile 0 and three half-tiles of K-tile 1, in the order the loop continues.  (The staged tile of
        // the previous round is no longer read: every wave passed the barrier behind its stores.)
        decode(0);
        issue_p(0, 0);
        issue_w(0, 0, 0);
        issue_w(1, 1, 0);
        issue_p(1, 1);
        decode(1);
        issue_p(0, 2);
        issue_w(0, 2, 1);
        issue_p(1, 3);
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(VM) : "memory");  // K-tile 0 has landed for every wave
        if (late) PH_BARRIER();

        for (int kt = 0; kt < nk; ++kt) {
            const int s0 = (2 * kt) & 3, s1 = s0 + 1;
            // phase 0: quadrant (P0, W0); refill W1 of K-tile kt+1
            PH_READ_P(s0, MF0)
            PH_READ_W(w0, s0, NF0)
            issue_w(1, (s1 + 2) & 3, kt + 1);
            PH_SYNC_READS();
            PH_MMA(0, MF0, 0, NF0, w0)
            PH_BARRIER();
            // phase 1: quadrant (P0, W1); refill P0 of K-tile kt+2 (the slot of this K-tile's P0, read in phase 0)
            PH_READ_W(w1, s1, NF1)
            decode(kt + 2);
            issue_p(0, s0);
            PH_SYNC_READS();
            PH_MMA(0, MF0, NF0, NF1, w1)
            PH_BARRIER();
            // phase 2: quadrant (P1, W1); refill W0 of K-tile kt+2
            PH_READ_P(s1, MF1)
            issue_w(0, s0, kt + 2);
            PH_SYNC_READS();
            PH_MMA(MF0, MF1, NF0, NF1, w1)
            PH_BARRIER();
            // phase 3: quadrant (P1, W0), W0 still in registers; refill P1 of K-tile kt+2; K-tile kt+1 must have landed
            issue_p(1, s1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM) : "memory");
            PH_SYNC_READS();
            PH_MMA(MF0, MF1, 0, NF0, w0)
            PH_BARRIER();
        }
        if (!late) PH_BARRIER();
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // the zero fills of the last K-tiles are in: the ring is free

        // ---- epilogue: post-ops in registers -> 16-bit tile in LDS -> 16-byte stores (+ residual)
        PH_ARGS(a)
        const bool has_bias = a.bias != nullptr, has_scale = a.scale != nullptr;
        epi_store(prm, epi, t, BN);
        PH_SYNC_READS();
        // accumulator (mi, nj), element e: pixel wm*MF*16 + mi*16 + lr, channel wn*NF*16 + nj*16 + lq*4 + e
#pragma unroll
        for (int nj = 0; nj < NF; ++nj) {
            const int nl = wn * (NF * 16) + nj * 16 + lq * 4;
            if (n0 + nl < a.Cout) {
#pragma unroll
                for (int mi = 0; mi < MF; ++mi) {
                    const int ml = wm * (MF * 16) + mi * 16 + lr;
                    *reinterpret_cast<ev4*>(smem + ml * EP + nl * 2) = post_op4_lds<ET>(acc[mi][nj], nl, prm, BN, has_bias, has_scale, a.slope);
                }
            }
        }
        PH_SYNC_READS();
        constexpr int CPR = BN / 8;  // 16-byte chunks per tile row
        ET* y = reinterpret_cast<ET*>(a.y);
        const ET* res = reinterpret_cast<const ET*>(a.res);
#pragma unroll 4
        for (int idx = t; idx < BM * CPR; idx += 512) {
            const int row = idx / CPR, cc = idx - row * CPR;
            const int m = m0 + row, n = n0 + cc * 8;
            if (m < a.M && n < a.Cout) {
                ev8 v = *reinterpret_cast<const ev8*>(smem + row * EP + cc * 16);
                if (res) v = Elem<ET>::add(v, *reinterpret_cast<const ev8*>(res + (size_t)m * a.res_ld + n));
                *reinterpret_cast<ev8*>(y + (size_t)m * a.y_ld + n) = v;
            }
        }
        PH_SYNC_READS();  // every wave has read its part of the staged tile: the next round's DMAs may overwrite it
    }
}
#undef PH_READ_P
#undef PH_READ_W
#undef PH_MMA
#undef PH_ARGS

int ph_cu_count() {
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

template <typename ET, int WM, int WN, int MF, int NF>
int launch_ph(const ConvArgs& a0, hipStream_t s) {
    typedef PhCfg<WM, WN, MF, NF> Cfg;
    ConvArgs a = a0;
    const long long tiles_m = (a.M + Cfg::BM - 1) / Cfg::BM;
    a.tiles_n = (a.Cout + Cfg::BN - 1) / Cfg::BN;
    a.nk = (a.K + 63) / 64;
    const long long total = tiles_m * a.tiles_n;
    if (total <= 0 || total > 0x7fffffffLL) return y5_fail(Y5_EINVAL, "conv: tile count %lld out of range", total);
    a.tiles_total = (int)total;
    static std::atomic<unsigned long long> attr_set{0};  // one bit per device: the attribute is per function and device
    const unsigned long long bit = 1ull << y5_current_device();
    if (!(attr_set.load() & bit)) {
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_ph_kernel<ET, WM, WN, MF, NF>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
        attr_set.fetch_or(bit);
    }
    const long long cus = ph_cu_count();
    const unsigned grid = (unsigned)(total < cus ? total : cus);  // one resident workgroup per compute unit, walking its tiles
    hipLaunchKernelGGL((conv_ph_kernel<ET, WM, WN, MF, NF>), dim3(grid), dim3(512), Cfg::SMEM, s, a);
    Y5_LAUNCH_CHECK("conv_ph_kernel");
    return Y5_OK;
}

struct PhVariant {
    int (*f16_launch)(const ConvArgs&, hipStream_t);
    int (*bf16_launch)(const ConvArgs&, hipStream_t);
};
#define Y5_PH(WM, WN, MF, NF) {&launch_ph<f16, WM, WN, MF, NF>, &launch_ph<bf16, WM, WN, MF, NF>}
const PhVariant kPh[] = {
    Y5_PH(2, 4, 8, 4),  // 256 pixels x 256 channels
    Y5_PH(4, 2, 8, 4),  // 512 x 128
    Y5_PH(2, 4, 7, 4),  // 224 x 256
    Y5_PH(4, 2, 5, 4),  // 320 x 128
    Y5_PH(4, 2, 5, 5),  // 320 x 160
    Y5_PH(2, 4, 6, 5),  // 192 x 320
};
#undef Y5_PH

}  // namespace

int y5_conv_ph_count() { return (int)(sizeof(kPh) / sizeof(kPh[0])); }

// which = 1..y5_conv_ph_count().  Returns 0 when the shape is outside what this kernel family covers.
int y5_try_conv_ph(const y5_conv_desc& d, hipStream_t s, int which, int* rc) {
    if (which < 1 || which > y5_conv_ph_count()) return 0;
    if (d.ksize != 1 && d.ksize != 3) return 0;
    if ((long long)d.w_rows * d.kpad * 2 >= 0x7fffffffLL) return 0;
    const long long img_span = (long long)d.h * d.w_in * d.x_ld * 2;
    if (img_span >= 0x7fffffffLL) return 0;
    // The loaders address x by 32-bit byte offsets: a batch whose input spans 2 GiB or more is convolved in slices of
    // whole images.
    const int per = (int)(0x7fffffffLL / img_span) < d.batch ? (int)(0x7fffffffLL / img_span) : d.batch;
    const bool bf = d.dtype == Y5_BF16;
    *rc = Y5_OK;
    for (int b0 = 0; b0 < d.batch && *rc == Y5_OK; b0 += per) {
        const int nb = d.batch - b0 < per ? d.batch - b0 : per;
        ConvArgs a{};
        a.x = static_cast<const f16*>(d.x) + (size_t)b0 * d.h * d.w_in * d.x_ld;
        a.w = static_cast<const f16*>(d.w);
        a.bias = d.bias;
        a.scale = d.scale;
        a.shift = d.shift;
        a.res = d.res ? static_cast<const f16*>(d.res) + (size_t)b0 * d.ho * d.wo * d.res_ld : nullptr;
        a.y = static_cast<f16*>(d.y) + (size_t)b0 * d.ho * d.wo * d.y_ld;
        a.M = (long long)nb * d.ho * d.wo;
        a.H = d.h; a.W = d.w_in; a.Cin = d.cin; a.x_ld = d.x_ld;
        a.Ho = d.ho; a.Wo = d.wo; a.Cout = d.cout; a.y_ld = d.y_ld; a.res_ld = d.res_ld;
        a.stride = d.stride; a.pad = d.pad; a.K = d.ksize * d.ksize * d.cin;
        a.w_ld = d.kpad;
        a.slope = d.act == Y5_ACT_LEAKY ? d.slope : 1.f;
        a.div_hw = make_fastdiv((unsigned)(d.ho * d.wo));
        a.div_wo = make_fastdiv((unsigned)d.wo);
        a.div_cin = make_fastdiv((unsigned)d.cin);
        a.x_bytes = (int)((((long long)nb * d.h * d.w_in - 1) * d.x_ld + d.cin) * 2);
        a.w_bytes = (int)((long long)d.w_rows * d.kpad * 2);
        *rc = bf ? kPh[which - 1].bf16_launch(a, s) : kPh[which - 1].f16_launch(a, s);
    }
    return 1;
}
