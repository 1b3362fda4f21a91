// Phased implicit-GEMM convolution ("ph"): the kernel for the compute-bound layers (3x3 / 1x1, any stride of {1, 2},
// any channel counts that are multiples of 8; what models/yolo.py:195 and structured pruning produce), fp16 or bf16.
//
// GEMM view as in conv_igemm.hip: D[n][m] = sum_k W[n][k] * X[m][k], m = output pixel, n = output channel,
// k = (tap, input channel) flattened - a 64-wide K-tile may span taps, so K = 9 * Cin is walked without per-tap padding
// (Cin = 80 costs 12 K-tiles, not 18).  What differs is the schedule:
//   * eight waves as WM x WN, each owning MF x NF accumulators of v_mfma_f32_16x16x32 (up to 128 pixels x 64 channels: four
//     fragment reads per eight multiplies); the compiled tiles run from 256 x 256 for wide layers to 512 x 80 for 80-channel ones;
//   * a K-tile is staged as four half-tiles (pixel halves P0 / P1 = the rows of the waves' first / second MF halves,
//     weight halves W0 / W1 likewise for NF) into a ring of four slots, by buffer_load ... lds issued by all waves, up to
//     1.5 K-tiles ahead; 128-byte LDS rows are XOR-swizzled through the copies' source addresses;
//   * a K-tile is multiplied in two phases (the accumulators of the first / second pixel half): fragment reads of the phase,
//     the copies of one or three half-tiles, [barrier], the phase's MFMAs, [barrier].  Waves 4-7 run one barrier behind waves
//     0-3, so on every SIMD one wave multiplies while its partner reads LDS and issues copies;
//   * the only wait for copies is a counted s_waitcnt vmcnt(N) that leaves the youngest half-tiles in flight; barriers are
//     LDS-only;
//   * workgroups are persistent (one per compute unit, tiles in XCD-contiguous order); with Cin a multiple of 64 a K-tile
//     lies in one tap and its tap offset and validity bit are scalar (UNI);
//   * epilogue: post-ops in registers -> 16-bit tile in LDS -> 16-byte stores, the residual values fetched in batches of four
//     chunks ahead of the store loop.
// Hazards.  RAW: a half-tile is read no earlier than the phase after the counted wait that retired it, behind a barrier
// every wave has passed after its own wait.  WAR: a slot is refilled one phase after its last read, and every phase
// drains its reads (lgkmcnt(0)) before its first barrier.  K-tiles past the end are "issued" with out-of-range offsets
// (the descriptor's range check turns them into zero fills of free slots), which keeps the wait count constant.
#include <type_traits>

#include "y5_common.h"

namespace {

#include "conv_common.h"

// dma16_buf (conv_common.h) leaving M0 set to the copy's LDS address instead of restoring it: two scalar moves fewer per copy where copy
// issue is what a kernel's phases wait for (the compiler is told that M0 is overwritten).
__device__ __forceinline__ void dma16_buf_m0(i32x4 rsrc, unsigned voff, unsigned lds_base) {
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %0, %1, 0 offen lds"
        :
        : "v"(voff), "s"(rsrc), "s"(lds_base)
        : "memory", "m0");
}


// WM x WN waves (eight), each owning MF x NF accumulators of TS pixels x TS channels (TS = 16: v_mfma_f32_16x16x32, two per
// 64-wide K-tile and accumulator; TS = 32: v_mfma_f32_32x32x16, four): tile = (WM*MF*TS) pixels x (WN*NF*TS) channels.
// The 32 x 32 form holds a SIMD's instruction issue for 8 of its 32 cycles, the 16 x 16 form for 8 of 16: beside a partner
// wave that has ~60 address / read / copy instructions to issue per phase, the wide form leaves it the slots it needs.  MF and NF pick the tile that fills the chip in whole rounds for a layer's pixel and channel
// counts (the engine times the compiled shapes per layer); the halves of the schedule hold ceil / floor of them.
template <int WM, int WN, int MF, int NF, int TS>
struct PhCfg {
    static_assert(WM * WN == 8, "eight waves");
    static_assert(TS == 16 || TS == 32, "MFMA tile");
    static_assert(MF >= 2 && NF >= 2 && MF * NF * TS * TS <= 32 * 256, "at most 128 accumulator registers");
    static constexpr int MF0 = (MF + 1) / 2, MF1 = MF / 2, NF0 = (NF + 1) / 2, NF1 = NF / 2;
    static constexpr int BM = WM * MF * TS, BN = WN * NF * TS;
    static constexpr int PR = MF0 * TS, WR = NF0 * TS;            // rows a wave row / wave column owns in a slot
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

#ifdef Y5_STAMPS
// Diagnostic build only (make dbg): per workgroup and tile, wave 0: {start, set-up done, K-tile 0 landed, K loop end, tile
// staged, stores issued} in shader cycles, then the 100 MHz real-time counter at start and end
__device__ unsigned long long y5_ph_stamp_buf[8 * 65536];
#define PH_STAMP(I) if (t == 0 && stamp_row < 65536) y5_ph_stamp_buf[8 * stamp_row + (I)] = __builtin_amdgcn_s_memtime();
// K-loop sections of waves 0 and 4, summed over the K-tiles of a tile: {issue + reads, vmcnt wait, lgkmcnt wait + barrier,
// MFMAs, second barrier}
__device__ unsigned long long y5_ph_sect_buf[2 * 8 * 65536];
#define PH_SECT(I) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); sect[I] += now_ - sect_t; sect_t = now_; }
#define PH_RSTAMP(I) if (t == 0 && stamp_row < 65536) y5_ph_stamp_buf[8 * stamp_row + (I)] = __builtin_amdgcn_s_memrealtime();
#else
#define PH_STAMP(I)
#define PH_RSTAMP(I)
#define PH_SECT(I)
#endif

#define PH_SYNC_READS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define PH_BARRIER() asm volatile("s_barrier" ::: "memory")

// UNI: Cin is a multiple of 64, so a K-tile lies in one tap: its tap, validity bit and source offset are wave-uniform
// (scalar arithmetic) instead of per lane.
template <typename ET, int WM, int WN, int MF, int NF, int TS, bool UNI>
__global__ __launch_bounds__(512, 2) void conv_ph_kernel(const ConvArgs a_unused) {
    typedef PhCfg<WM, WN, MF, NF, TS> Cfg;
    constexpr int KSN = TS == 16 ? 2 : 4;   // MFMA K-steps per 64-wide K-tile
    constexpr int FB = TS * 128;            // bytes of one fragment's rows in a slot
    typedef typename std::conditional<TS == 16, f32x4, f32x16>::type accv;
    typedef typename Elem<ET>::v8 ev8;
    typedef typename Elem<ET>::v4 ev4;
    constexpr int MF0 = Cfg::MF0, MF1 = Cfg::MF1, NF0 = Cfg::NF0, NF1 = Cfg::NF1;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, EP = Cfg::EP, PR = Cfg::PR, WR = Cfg::WR;
    constexpr int P_SLOT = Cfg::P_SLOT, W_SLOT = Cfg::W_SLOT, W_RING = 4 * P_SLOT;
    constexpr int NP = Cfg::NP, NW = Cfg::NW;
    constexpr int VM = 2 * NP + 2 * NW;       // four half-tiles (P1, P0, W0, W1) stay in flight at the wait of a phase
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
    // (TS = 32: row l & 31, K-chunk 2*ks + (l >> 5))
    const int lr = lane & (TS - 1), lq = lane / TS;
    const int sw = (lr >> 1) & 7;  // f(row): the groups start at multiples of 16 rows
    int coff[KSN];
#pragma unroll
    for (int ks = 0; ks < KSN; ++ks) coff[ks] = ((((64 / TS) * ks + lq) ^ sw) << 4);
    const unsigned char* prow = smem + (wm * PR + lr) * 128;
    const unsigned char* wrow = smem + W_RING + (wn * WR + lr) * 128;
    float* prm = reinterpret_cast<float*>(smem + Cfg::TILE);

#define PH_READ_P(SLOT, N)                                                                           \
    _Pragma("unroll") for (int i = 0; i < (N); ++i)                                                  \
        _Pragma("unroll") for (int ks = 0; ks < KSN; ++ks)                                           \
            pf[i][ks] = *reinterpret_cast<const ev8*>(prow + (SLOT) * P_SLOT + i * FB + coff[ks]);
#define PH_READ_W(J0, SLOT, N)                                                                       \
    _Pragma("unroll") for (int j = 0; j < (N); ++j)                                                  \
        _Pragma("unroll") for (int ks = 0; ks < KSN; ++ks)                                           \
            wf[(J0) + j][ks] = *reinterpret_cast<const ev8*>(wrow + (SLOT) * W_SLOT + j * FB + coff[ks]);
#define PH_MMA(M0, MN)                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    __builtin_amdgcn_s_setprio(1);                                                                   \
    _Pragma("unroll") for (int ks = 0; ks < KSN; ++ks)                                               \
        _Pragma("unroll") for (int i = 0; i < (MN); ++i)                                             \
            _Pragma("unroll") for (int j = 0; j < NF; ++j)                                           \
                acc[(M0) + i][j] = mma(wf[j][ks], pf[i][ks], acc[(M0) + i][j]);                      \
    __builtin_amdgcn_s_setprio(0);                                                                   \
    __builtin_amdgcn_sched_barrier(0);

    auto mma = [](ev8 w, ev8 p, accv c) -> accv {
        if constexpr (TS == 16) return Mma16<ET>::run(w, p, c);
        else return Elem<ET>::mfma(w, p, c);
    };
    // persistent: a workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...
    for (int it = blockIdx.x; it < tiles_total; it += gridDim.x) {
        const int tile = xcd_remap(it, tiles_total);
        const int mt = tile / tiles_n, nt = tile - mt * tiles_n;
        const int m0 = mt * BM, n0 = nt * BN;
#ifdef Y5_STAMPS
        const int stamp_row = (int)blockIdx.x + (int)gridDim.x * ((it - (int)blockIdx.x) / (int)gridDim.x);
#endif
        PH_STAMP(0)
        PH_RSTAMP(6)
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
                const int m = m0 + wrow_ * (MF * TS) + h * (MF0 * TS) + within;  // its pixel
                const bool mvalid = pin && within < (h ? MF1 : MF0) * TS && m < a.M;
                const unsigned mm = mvalid ? (unsigned)m : 0u;
                const unsigned b = fdiv(mm, div_hw);
                const unsigned rem = mm - b * (unsigned)(a.Ho * a.Wo);
                const unsigned oy = fdiv(rem, div_wo), ox = rem - oy * (unsigned)a.Wo;
                const int hi0 = (int)oy * a.stride - a.pad, wi0 = (int)ox * a.stride - a.pad;
                pbase[h][i] = (((int)b * a.H + hi0) * a.W + wi0) * a.x_ld * 2 + (UNI ? kc * 16 : 0);
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
                const int n = n0 + wcol * (NF * TS) + h * (NF0 * TS) + within;
                const bool nvalid = pin && within < (h ? NF1 : NF0) * TS;
                // rows past the packed weights fail the range check like the marker does: zeros
                wbase[h][i] = nvalid ? ((unsigned)n * (unsigned)a.w_ld + kc * 8) * 2u : Y5_BUF_OOB;
            }
        }
        }
        // position of this thread's chunk in K-tile kt: 1 << tap (0 when the chunk lies past K or the K-tile past the
        // last one) and the byte offset of (tap, channel) from tap (0, 0); with UNI both are the same for all lanes (the
        // lane's K-chunk is then part of pbase)
        unsigned pbit = 0;
        int poff = 0;
        auto decode = [&](int kt) {
            if constexpr (UNI) {
                const unsigned k = (unsigned)kt * 64u;
                const unsigned tp = (unsigned)(((unsigned long long)k * cin_mul) >> 32) >> cin_shr;
                const int c = (int)(k - tp * argCin);
                const int r = (int)(tp * 11u) >> 5, q = (int)tp - 3 * r;
                poff = ((r * argW + q) * arg_xld + c) * 2;
                pbit = kt < nk ? 1u << (tp & 15u) : 0u;
            } else {
                const unsigned k = (unsigned)kt * 64u + (unsigned)kc * 8u;
                const unsigned tp = __umulhi(k, cin_mul) >> cin_shr;  // k / Cin (Cin >= 8: never the divisor-1 case)
                const int c = (int)(k - tp * argCin);
                const int r = (int)(tp * 11u) >> 5, q = (int)tp - 3 * r;
                poff = ((r * argW + q) * arg_xld + c) * 2;
                pbit = (kt < nk && k < argK) ? 1u << (tp & 15u) : 0u;  // k < K: tap < 9
            }
        };
        auto issue_p = [&](int h, int slot) {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const bool out = 8 * i + 7 >= Cfg::P_PIECES && wave + 8 * i >= Cfg::P_PIECES;  // only a slot's last pieces can lie past it
                const unsigned dst = out ? (unsigned)Cfg::RING : (unsigned)(slot * P_SLOT + (wave + 8 * i) * 1024);
                dma16_buf_m0(xr, (pmask[i] & (pbit << (16 * h))) ? (unsigned)(pbase[h][i] + poff) : Y5_BUF_OOB, lds0 + dst);
            }
        };
        auto issue_w = [&](int h, int slot, int kt) {
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const bool out = 8 * i + 7 >= Cfg::W_PIECES && wave + 8 * i >= Cfg::W_PIECES;
                const unsigned dst = out ? (unsigned)Cfg::RING : (unsigned)(W_RING + slot * W_SLOT + (wave + 8 * i) * 1024);
                // past the last K-tile: + 2^30 puts a valid row offset (the packed weights span less than 2^30 bytes) past
                // the descriptor's range and keeps the marker of an unused row there (no wrap)
                dma16_buf_m0(wr, wbase[h][i] + (kt < nk ? (unsigned)kt * 128u : 0x40000000u), lds0 + dst);
            }
        };

        ev8 pf[MF0][KSN], wf[NF][KSN];
        accv acc[MF][NF];
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j)
#pragma unroll
                for (int e = 0; e < TS * TS / 64; ++e) acc[i][j][e] = 0.f;

        // ---- prologue: K-tile 0 and P0, W0, W1 of K-tile 1, in the order the loop continues.  (The staged tile of the
        // previous round is no longer read: every wave passed the barrier behind its stores.)
        PH_STAMP(1)
        decode(0);
        issue_p(0, 0);
        issue_w(0, 0, 0);
        issue_w(1, 1, 0);
        issue_p(1, 1);
        decode(1);
        issue_p(0, 2);
        issue_w(0, 2, 1);
        issue_w(1, 3, 1);
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NP + 2 * NW) : "memory");  // K-tile 0 has landed for every wave
        if (late) PH_BARRIER();
        PH_STAMP(2)

        // Two phases per K-tile, each one half of the wave's pixels by all of its channels.  Copies in issue order:
        // ... [B of kt-1: P0, W0, W1 of kt+1] [A of kt: P1 of kt+1] [B of kt: P0, W0, W1 of kt+2] ...; every phase waits
        // until all but the youngest VM copies are in, which retires what the NEXT phase reads; a slot is refilled no
        // earlier than the phase after its last read, and every phase drains its reads before its first barrier.
#ifdef Y5_STAMPS
        unsigned long long sect[5] = {0, 0, 0, 0, 0}, sect_t = __builtin_amdgcn_s_memtime();
#endif
        for (int kt = 0; kt < nk; ++kt) {
            const int s0 = (2 * kt) & 3, s1 = s0 + 1;
            // phase A: pixels P0 x channels W0 | W1; P1 of K-tile kt+1 goes where P1 of kt-1 was
            PH_READ_P(s0, MF0)
            PH_READ_W(0, s0, NF0)
            PH_READ_W(NF0, s1, NF1)
            issue_p(1, (s1 + 2) & 3);
            PH_SECT(0)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM) : "memory");
            PH_SECT(1)
            PH_SYNC_READS();
            PH_SECT(2)
            PH_MMA(0, MF0)
            PH_SECT(3)
            PH_BARRIER();
            PH_SECT(4)
            // phase B: pixels P1, the weights still in registers; P0, W0, W1 of K-tile kt+2 go where this K-tile's were
            PH_READ_P(s1, MF1)
            decode(kt + 2);
            issue_p(0, s0);
            issue_w(0, s0, kt + 2);
            issue_w(1, s1, kt + 2);  // (phase A of kt+1 issues P1 of kt+2 with this decode)
            PH_SECT(0)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM) : "memory");
            PH_SECT(1)
            PH_SYNC_READS();
            PH_SECT(2)
            PH_MMA(MF0, MF1)
            PH_SECT(3)
            PH_BARRIER();
            PH_SECT(4)
        }
#ifdef Y5_STAMPS
        if ((t == 0 || t == 256) && stamp_row < 65536)
            for (int i = 0; i < 5; ++i) y5_ph_sect_buf[(2 * stamp_row + (t >> 8)) * 8 + i] = sect[i];
#endif
        if (!late) PH_BARRIER();
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // the zero fills of the last K-tiles are in: the ring is free
        PH_STAMP(3)

        // ---- epilogue: post-ops in registers -> 16-bit tile in LDS -> 16-byte stores (+ residual).  The residual values
        // are fetched first, into the registers the fragments have left, so that their latency lies under the post-ops.
        PH_ARGS(a)
        const bool has_bias = a.bias != nullptr, has_scale = a.scale != nullptr;
        constexpr int CPR = BN / 8;                      // 16-byte chunks per tile row
        constexpr int ITER = (BM * CPR + 511) / 512;     // chunks per thread
        ET* y = reinterpret_cast<ET*>(a.y);
        const ET* res = reinterpret_cast<const ET*>(a.res);
        constexpr int NB = 4;                            // chunks per batch: one batch of residual values in flight
        ev8 rv[NB];
        auto fetch_res = [&](int i0) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int idx = t + (i0 + j) * 512;
                const int row = idx / CPR, cc = idx - row * CPR;
                const int m = m0 + row, n = n0 + cc * 8;
                if (idx < BM * CPR && m < a.M && n < a.Cout) rv[j] = *reinterpret_cast<const ev8*>(res + (size_t)m * a.res_ld + n);
            }
        };
        epi_store(prm, epi, t, BN);
        PH_SYNC_READS();
        // accumulator (mi, nj): pixel wm*MF*TS + mi*TS + lr; TS = 16: element e is channel wn*NF*16 + nj*16 + 4*lq + e;
        // TS = 32: elements 4g .. 4g+3 are channels wn*NF*32 + nj*32 + 8g + 4*lq .. + 3
#pragma unroll
        for (int nj = 0; nj < NF; ++nj) {
#pragma unroll
            for (int g = 0; g < TS / 16 * (TS / 16); ++g) {
                const int nl = wn * (NF * TS) + nj * TS + (TS == 16 ? lq * 4 : 8 * g + 4 * lq);
                if (n0 + nl < a.Cout) {
#pragma unroll
                    for (int mi = 0; mi < MF; ++mi) {
                        const int ml = wm * (MF * TS) + mi * TS + lr;
                        const f32x4 v = {acc[mi][nj][4 * g], acc[mi][nj][4 * g + 1], acc[mi][nj][4 * g + 2], acc[mi][nj][4 * g + 3]};
                        *reinterpret_cast<ev4*>(smem + ml * EP + nl * 2) = post_op4_lds<ET>(v, nl, prm, BN, has_bias, has_scale, a.slope);
                    }
                }
            }
        }
        if (res) fetch_res(0);
        PH_SYNC_READS();
        PH_STAMP(4)
#pragma unroll 1
        for (int i0 = 0; i0 < ITER; i0 += NB) {
            ev8 cur[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) cur[j] = rv[j];
            if (res && i0 + NB < ITER) fetch_res(i0 + NB);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int idx = t + (i0 + j) * 512;
                const int row = idx / CPR, cc = idx - row * CPR;
                const int m = m0 + row, n = n0 + cc * 8;
                if (idx < BM * CPR && m < a.M && n < a.Cout) {
                    ev8 v = *reinterpret_cast<const ev8*>(smem + row * EP + cc * 16);
                    if (res) v = Elem<ET>::add(v, cur[j]);
                    *reinterpret_cast<ev8*>(y + (size_t)m * a.y_ld + n) = v;
                }
            }
        }
        PH_SYNC_READS();  // every wave has read its part of the staged tile: the next round's DMAs may overwrite it
        PH_STAMP(5)
        PH_RSTAMP(7)
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

template <typename ET, int WM, int WN, int MF, int NF, int TS>
int launch_ph(const ConvArgs& a0, hipStream_t s) {
    const bool uni = a0.Cin % 64 == 0;
    typedef PhCfg<WM, WN, MF, NF, TS> Cfg;
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
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_ph_kernel<ET, WM, WN, MF, NF, TS, false>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_ph_kernel<ET, WM, WN, MF, NF, TS, true>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
        attr_set.fetch_or(bit);
    }
    const long long cus = ph_cu_count();
    const unsigned grid = (unsigned)(total < cus ? total : cus);  // one resident workgroup per compute unit, walking its tiles
    if (uni)
        hipLaunchKernelGGL((conv_ph_kernel<ET, WM, WN, MF, NF, TS, true>), dim3(grid), dim3(512), Cfg::SMEM, s, a);
    else
        hipLaunchKernelGGL((conv_ph_kernel<ET, WM, WN, MF, NF, TS, false>), dim3(grid), dim3(512), Cfg::SMEM, s, a);
    Y5_LAUNCH_CHECK("conv_ph_kernel");
    return Y5_OK;
}

struct PhVariant {
    int (*f16_launch)(const ConvArgs&, hipStream_t);
    int (*bf16_launch)(const ConvArgs&, hipStream_t);
};
#define Y5_PH(WM, WN, MF, NF, TS) {&launch_ph<f16, WM, WN, MF, NF, TS>, &launch_ph<bf16, WM, WN, MF, NF, TS>}
const PhVariant kPh[] = {
    Y5_PH(2, 4, 8, 4, 16),  // 256 pixels x 256 channels
    Y5_PH(4, 2, 8, 4, 16),  // 512 x 128
    Y5_PH(2, 4, 7, 4, 16),  // 224 x 256
    Y5_PH(4, 2, 5, 4, 16),  // 320 x 128
    Y5_PH(4, 2, 5, 5, 16),  // 320 x 160
    Y5_PH(8, 1, 4, 5, 16),  // 512 x 80
    Y5_PH(2, 4, 8, 3, 16),  // 256 x 192
    Y5_PH(4, 2, 8, 3, 16),  // 512 x 96
};
#undef Y5_PH

}  // namespace

#ifdef Y5_STAMPS
extern "C" int y5_debug_sections_ph(unsigned long long* dst, int rows) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(y5_ph_sect_buf), (size_t)rows * 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
extern "C" int y5_debug_stamps_ph(unsigned long long* dst, int rows) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(y5_ph_stamp_buf), (size_t)rows * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif

int y5_conv_ph_count() { return (int)(sizeof(kPh) / sizeof(kPh[0])); }

// which = 1..y5_conv_ph_count().  Returns 0 when the shape is outside what this kernel family covers.
int y5_try_conv_ph(const y5_conv_desc& d, hipStream_t s, int which, int* rc) {
    if (which < 1 || which > y5_conv_ph_count()) return 0;
    if (d.ksize != 1 && d.ksize != 3) return 0;
    if ((long long)d.w_rows * d.kpad * 2 >= 0x40000000LL) return 0;
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
