// Implicit-GEMM convolution on MFMA for gfx950: NHWC fp16 activations, fp32 accumulation,
// fused bias / folded-BatchNorm / LeakyReLU / residual epilogue, output into a channel slice.
//
// GEMM view:  D[n][m] = sum_k Wt[n][k] * X[m][k]
//   m = (b, oy, ox) output pixel, n = output channel, k = (r*ks + q)*cin + c.
// One workgroup of 4 or 8 waves owns a BM-pixel x BN-channel tile and walks K in steps of BK
// (32 or 64): the X tile is gathered (im2col on the fly, a zero page for padding) and the Wt tile copied
// by 16-byte global->LDS DMA into an S-deep ring of LDS stages, S-1 steps ahead of the MFMAs, with
// hand-counted s_waitcnt vmcnt and one barrier per step.  The LDS image is lane-linear per DMA
// instruction; a 16-byte XOR swizzle (applied to the source chunk and to the read address) keeps the
// ds_read_b128 fragment reads bank-conflict free.  v_mfma_f32_32x32x16_f16 takes the weights as the
// row operand, so every lane ends up holding 4 consecutive channels of one pixel per accumulator
// group.  The epilogue applies the per-channel post-ops on those groups of four with the reference's
// fp16 rounding points, stages the tile in LDS and writes whole 16-byte channel vectors, adding the
// residual there.
#include <cstdlib>

#include "y5_common.h"

namespace {

#include "conv_common.h"

#ifdef Y5_STAMPS
// Diagnostic build only (make dbg): per-wave cycle totals of the K loop's phases, read back by y5_debug_stamps().
__device__ unsigned long long y5_stamp_buf[8 * 65536];
#define Y5_STAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define Y5_STAMP(v)
#endif

// BM x BN output tile, WM x WN waves, KS in {1,3}, S-deep LDS ring of BK-wide K-steps filled by DMA S-1 steps
// ahead.  DIRECT: 1x1 stride-1 convolution, where output pixel m reads input pixel m (no coordinate decode).
template <int BM, int BN, int WM, int WN, int S, int BK>
struct IgemmCfg {
    static constexpr int NT = 64 * WM * WN;
    static constexpr int STAGE_BYTES = (BM + BN) * BK * 2;
    static constexpr int EP = BN * 2 + 16;  // epilogue row pitch in bytes
    static constexpr int PRM0 = (S * STAGE_BYTES > BM * EP) ? S * STAGE_BYTES : BM * EP;  // epilogue parameters behind ring / tile
    static constexpr int SMEM = PRM0 + 3 * BN * 4;
};

// UT ("uniform tap", KS = 3): K-steps restart at every tap, so every chunk of a K-step belongs to the same tap, the tap's
// pixel offset is one wave-uniform number and a thread only keeps, per staged row, the offset of its pixel at tap
// (0, 0) and nine validity bits -- a DMA instruction then costs ~8 instructions of address work instead of ~20.  Where
// Cin is not a multiple of BK the tap's last step stages zeros for the missing channels (and whatever weights follow in
// the row: their products vanish).
// CAT2 (1x1, DIRECT): the input is the channel concatenation [nearest-2x-upsampled x | x2] that is never built: K-steps
// below Cin (a multiple of BK) fetch pixel (oy/2, ox/2) of the half-resolution x, the rest pixel (oy, ox) of x2.
// ET: element type of activations and weights (f16 or bf16); everything up to the MFMA and from the post-ops on is
// 16-bit data movement that does not care.
template <int BM, int BN, int WM, int WN, int KS, int S, int BK, bool DIRECT, bool UT = false, bool CAT2 = false, typename ET = f16>
__global__ __launch_bounds__(64 * WM * WN) void conv_igemm_kernel(const ConvArgs a) {
    typedef Elem<ET> E;
    typedef typename E::v4 ev4;
    typedef typename E::v8 ev8;
    static_assert(!UT || (KS == 3 && !DIRECT), "uniform-tap addressing is for 3x3 convolutions");
    static_assert(!CAT2 || DIRECT, "the two-source input is for 1x1 convolutions");
    constexpr int NW = WM * WN, NT = 64 * NW;
    static_assert(NW == 4 || NW == 8, "four or eight waves per workgroup");
    static_assert(BK == 32 || BK == 64, "K-step is 32 or 64");
    static_assert(!DIRECT || KS == 1, "direct addressing is for 1x1 convolutions");
    constexpr int WTM = BM / WM, WTN = BN / WN;  // wave tile: pixels x channels
    constexpr int RM = WTM / 32, RN = WTN / 32;  // 32x32 MFMA tiles per wave
    static_assert(RM >= 1 && RN >= 1, "wave tile must hold at least one MFMA tile");
    constexpr int ROW_BYTES = BK * 2;            // LDS bytes per tile row and step
    constexpr int CPRW = BK / 8;                 // 16-byte chunks per tile row (4 or 8)
    constexpr int RPI = NT / CPRW;               // tile rows one DMA instruction of the workgroup covers
    constexpr int B_WAVES = BN * CPRW < NT ? BN * CPRW / 64 : NW;  // a weight tile smaller than one workgroup-wide DMA is
                                                                   // fetched by the first B_WAVES waves only
    static_assert(BM % RPI == 0 && (BN % RPI == 0 || (BN * CPRW < NT && BN * CPRW % 64 == 0)), "tile rows are staged RPI at a time");
    constexpr int A_CH = BM / RPI;               // DMA instructions per wave and K-step for activations
    constexpr int B_CH = (BN + RPI - 1) / RPI;   // ... for weights
    constexpr int PIECE = NT * 16;               // LDS bytes one workgroup-wide DMA instruction fills
    constexpr int STAGE_BYTES = IgemmCfg<BM, BN, WM, WN, S, BK>::STAGE_BYTES;
    constexpr int EP = IgemmCfg<BM, BN, WM, WN, S, BK>::EP;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave % WM, wn = wave / WM;

    const int tile = xcd_remap(blockIdx.x, a.tiles_total);
    const int mt = tile / a.tiles_n, nt = tile - mt * a.tiles_n;
    const int m0 = mt * BM;  // M < 2^31 is checked by the launcher
    const int n0 = nt * BN;
    float* prm = reinterpret_cast<float*>(smem + IgemmCfg<BM, BN, WM, WN, S, BK>::PRM0);
    const EpiParams epi = epi_fetch(a, n0, t, BN);

    // ---- per-thread gather state.  Thread t fills LDS slot (row = t/CPRW [+RPI*i], 16-byte slot t%CPRW) of
    // every stage; with the XOR swizzle that slot holds K-chunk kc of the row.
    const int row0 = t / CPRW;
    const int kc = (BK == 64) ? ((t & 7) ^ ((row0 >> 1) & 7)) : ((t & 3) ^ ((row0 >> 2) & 3));
    // Sources are byte offsets from a.x / a.w, fetched through buffer descriptors (conv_common.h: dma16_buf).
    const i32x4 xr = make_buf(a.x, a.x_bytes), wr = make_buf(a.w, a.w_bytes);
    const i32x4 xr2 = CAT2 ? make_buf(a.x2, a.x2_bytes) : xr;
    int hi0[A_CH], wi0[A_CH];
    int xrow[A_CH];  // DIRECT: offset of the pixel; otherwise of pixel (b, 0, 0)
    int acur2[CAT2 ? A_CH : 1];  // CAT2: running offset of the pixel in x2
    bool mvalid[A_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const int m = m0 + row0 + RPI * i;
        mvalid[i] = m < a.M;
        const unsigned mm = mvalid[i] ? (unsigned)m : 0u;
        if (CAT2) {
            const unsigned b = fdiv(mm, a.div_hw);
            const unsigned rem = mm - b * (unsigned)(a.Ho * a.Wo);
            const unsigned oy = fdiv(rem, a.div_wo), ox = rem - oy * (unsigned)a.Wo;
            hi0[i] = wi0[i] = 0;
            xrow[i] = (int)(((b * (unsigned)(a.Ho >> 1) + (oy >> 1)) * (unsigned)(a.Wo >> 1) + (ox >> 1)) * (unsigned)a.x_ld * 2u);
            acur2[i] = (int)(mm * (unsigned)a.x2_ld * 2u) + kc * 16;
        } else if (DIRECT) {
            hi0[i] = wi0[i] = 0;
            xrow[i] = (int)(mm * (unsigned)a.x_ld * 2u);
        } else {
            const unsigned b = fdiv(mm, a.div_hw);
            const unsigned rem = mm - b * (unsigned)(a.Ho * a.Wo);
            const unsigned oy = fdiv(rem, a.div_wo), ox = rem - oy * (unsigned)a.Wo;
            hi0[i] = (int)oy * a.stride - a.pad;
            wi0[i] = (int)ox * a.stride - a.pad;
            xrow[i] = (int)(b * (unsigned)(a.H * a.W) * (unsigned)a.x_ld * 2u);
        }
    }
    // UT / DIRECT: running source offset per staged row (UT: pixel of tap (0,0) -- possibly outside the image, never
    // used then -- plus this thread's chunk; DIRECT: the pixel's current K position) and validity bits
    int acur[A_CH];
    unsigned vmask[A_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        acur[i] = 0;
        vmask[i] = 0;
        if (UT) {
            acur[i] = xrow[i] + ((hi0[i] * a.W + wi0[i]) * a.x_ld + kc * 8) * 2;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int hi = hi0[i] + tp / 3, wi = wi0[i] + tp % 3;
                if (mvalid[i] && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W) vmask[i] |= 1u << tp;
            }
        } else if (DIRECT) {
            acur[i] = xrow[i] + kc * 16;
        }
    }
    int ut_tap = 0, ut_c0 = 0;  // wave-uniform: tap and first channel of the next K-step to fetch
    int ut_off = 0;             // ... and that tap's byte offset from tap (0, 0), its validity bit,
    unsigned ut_bit = 0;        // the step's byte offset in a weight row
    int ut_w = 0;

    const bool bload = wave < B_WAVES;  // wave-uniform
    unsigned wptr[B_CH];
#pragma unroll
    for (int i = 0; i < B_CH; ++i) wptr[i] = ((unsigned)(n0 + (bload ? row0 : 0) + RPI * i) * (unsigned)a.w_ld + kc * 8) * 2u;
    const int lps = A_CH + (bload ? B_CH : 0);  // DMA instructions this wave issues per K-step

    // running K position of this thread's chunk
    int kpos = kc * 8, tap = 0, cch = kpos;
    if (KS > 1) {
        tap = (int)fdiv((unsigned)kpos, a.div_cin);
        cch = kpos - tap * a.Cin;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

    // One K-step's staging is A_CH + B_CH DMA instructions per wave, issued back to back.  (Spreading them between the
    // MFMA groups of the step being multiplied was measured and gains nothing: the time goes into the DMA issue itself.)
    int st_r = 0, st_q = 0;
    bool st_kvalid = false;
    int k0u = 0;  // CAT2: first channel of the K-step being fetched (wave-uniform)
    auto step_begin = [&]() {
        if (UT) {
            const int r = (ut_tap * 11) >> 5, q = ut_tap - 3 * r;
            ut_off = ((r * a.W + q) * a.x_ld + ut_c0) * 2;
            ut_bit = 1u << ut_tap;
            ut_w = (ut_tap * a.Cin + ut_c0) * 2;
            return;
        }
        if (KS > 1) {
            st_r = (tap * 11) >> 5;  // tap / 3 for tap < 32
            st_q = tap - 3 * st_r;
        }
        st_kvalid = kpos < a.K;
    };
    auto issue_piece = [&](int p, int stage) {
        if (p < A_CH) {
            const int i = p;
            unsigned vo;
            if (UT) {
                vo = ((vmask[i] & ut_bit) && ut_c0 + kc * 8 < a.Cin) ? (unsigned)(acur[i] + ut_off) : Y5_BUF_OOB;
            } else if (CAT2) {
                if (k0u < a.Cin) {
                    vo = mvalid[i] ? (unsigned)acur[i] : Y5_BUF_OOB;
                    acur[i] += BK * 2;
                } else {
                    vo = (st_kvalid && mvalid[i]) ? (unsigned)acur2[i] : Y5_BUF_OOB;
                    acur2[i] += BK * 2;
                    dma16_buf(xr2, vo, lds0 + stage * STAGE_BYTES + wave * 1024 + i * PIECE);
                    return;
                }
            } else if (DIRECT) {
                vo = (st_kvalid && mvalid[i]) ? (unsigned)acur[i] : Y5_BUF_OOB;
                acur[i] += BK * 2;
            } else {
                const int hi = hi0[i] + st_r, wi = wi0[i] + st_q;
                const bool ok = st_kvalid && mvalid[i] && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
                vo = ok ? (unsigned)(xrow[i] + ((hi * a.W + wi) * a.x_ld + cch) * 2) : Y5_BUF_OOB;
            }
            dma16_buf(xr, vo, lds0 + stage * STAGE_BYTES + wave * 1024 + i * PIECE);
        } else if (bload) {
            const int i = p - A_CH;
            dma16_buf(wr, UT ? wptr[i] + (unsigned)ut_w : wptr[i], lds0 + stage * STAGE_BYTES + BM * ROW_BYTES + wave * 1024 + i * PIECE);
            if (!UT) wptr[i] += BK * 2;
        }
    };
    auto step_end = [&]() {
        if (UT) {
            ut_c0 += BK;
            if (ut_c0 >= a.Cin) {
                ut_c0 = 0;
                ++ut_tap;
            }
            return;
        }
        kpos += BK;
        if (CAT2) k0u += BK;
        if (KS > 1) {
            cch += BK;
            while (cch >= a.Cin) {
                cch -= a.Cin;
                ++tap;
            }
        } else {
            cch = kpos;
        }
    };
    auto issue = [&](int stage) {
        step_begin();
#pragma unroll
        for (int p = 0; p < A_CH + B_CH; ++p) issue_piece(p, stage);
        step_end();
    };

    f32x16 acc[RN][RM];
#pragma unroll
    for (int j = 0; j < RN; ++j)
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;

    const int lr = lane & 31, lh = lane >> 5;
    const int sw = (BK == 64) ? ((lr >> 1) & 7) : ((lr >> 2) & 3);  // wave tile rows start at multiples of 32
    const int nk = a.nk;

#pragma unroll
    for (int s = 0; s < S - 1; ++s)
        if (s < nk) issue(s);
    epi_store(prm, epi, t, BN);  // published by the barriers of the K loop
    const bool has_bias = a.bias != nullptr, has_scale = a.scale != nullptr;
#ifdef Y5_STAMPS
    unsigned long long st_wait = 0, st_issue = 0, st_mma = 0;
    const unsigned long long st_begin = __builtin_amdgcn_s_memtime();
#endif
    for (int kt = 0; kt < nk; ++kt) {
        // K-steps kt+1 .. min(kt+S-2, nk-1) may stay in flight; step kt must have landed for every wave
        int ahead = nk - 1 - kt;
        if (ahead > S - 2) ahead = S - 2;
        Y5_STAMP(s0);
        wait_vm_and_barrier(ahead * lps);
        Y5_STAMP(s1);
        if (kt + S - 1 < nk) issue((kt + S - 1) % S);
        Y5_STAMP(s2);
        const unsigned char* A = smem + (kt % S) * STAGE_BYTES;
        const unsigned char* B = A + BM * ROW_BYTES;
#ifdef Y5_STAMPS
        if (a.res_ld == -12345) continue;  // diagnostic: staging only, no LDS reads and no MFMAs
#endif
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int coff = ((2 * ks + lh) ^ sw) << 4;
            ev8 xf[RM], wf[RN];
#pragma unroll
            for (int i = 0; i < RM; ++i)
                xf[i] = *reinterpret_cast<const ev8*>(A + (wm * WTM + i * 32 + lr) * ROW_BYTES + coff);
#pragma unroll
            for (int j = 0; j < RN; ++j)
                wf[j] = *reinterpret_cast<const ev8*>(B + (wn * WTN + j * 32 + lr) * ROW_BYTES + coff);
#pragma unroll
            for (int j = 0; j < RN; ++j)
#pragma unroll
                for (int i = 0; i < RM; ++i)
                    acc[j][i] = E::mfma(wf[j], xf[i], acc[j][i]);
        }
#ifdef Y5_STAMPS
        const unsigned long long s3 = __builtin_amdgcn_s_memtime();
        st_wait += s1 - s0; st_issue += s2 - s1; st_mma += s3 - s2;
#endif
    }
    wait_vm_and_barrier(0);  // every wave is done reading the ring before the epilogue reuses it
#ifdef Y5_STAMPS
    const unsigned long long st_loop_end = __builtin_amdgcn_s_memtime();
#endif

    // ---- epilogue: post-ops in registers -> fp16 tile in LDS -> 16-byte stores (+ residual) ----
    // accumulator element e of MFMA tile (j,i): channel = 8*(e>>2) + 4*lh + (e&3) within the 32-channel
    // tile, pixel = lr; elements 4g..4g+3 are four consecutive channels.
#pragma unroll
    for (int j = 0; j < RN; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int nl = wn * WTN + j * 32 + 8 * g + 4 * lh;
            const int n = n0 + nl;
            if (n < a.Cout) {
#pragma unroll
                for (int i = 0; i < RM; ++i) {
                    const f32x4 v = {acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]};
                    const int ml = wm * WTM + i * 32 + lr;
                    *reinterpret_cast<ev4*>(smem + ml * EP + nl * 2) = post_op4_lds<ET>(v, nl, prm, BN, has_bias, has_scale, a.slope);
                }
            }
        }
    }
    __syncthreads();
    constexpr int CPR = BN / 8;          // 16-byte chunks per tile row
    constexpr int RPP = NT / CPR;        // rows per pass
    const int cc = t % CPR, rr = t / CPR;
    const int n = n0 + cc * 8;
    const bool want_stats = a.stats != nullptr;  // wave-uniform
    float ssum[8], ssq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) ssum[e] = ssq[e] = 0.f;
    if (n < a.Cout) {
#pragma unroll
        for (int p = 0; p < BM / RPP; ++p) {
            const int row = rr + p * RPP;
            const int m = m0 + row;
            if (m < a.M) {
                ev8 v = *reinterpret_cast<const ev8*>(smem + row * EP + cc * 16);
                if (a.res) v = E::add(v, *reinterpret_cast<const ev8*>(a.res + (size_t)m * a.res_ld + n));
                *reinterpret_cast<ev8*>(a.y + (size_t)m * a.y_ld + n) = v;
                if (want_stats) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float f = (float)v[e];
                        ssum[e] += f;
                        ssq[e] = __builtin_fmaf(f, f, ssq[e]);
                    }
                }
            }
        }
    }
    if (want_stats) {
        // per-channel sums of what was just stored: the RPP threads of a channel chunk meet in LDS (the staged tile is
        // dead once every thread has read its rows), one atomic per channel and moment leaves the workgroup
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);  // [RPP][CPR][16]
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[(rr * CPR + cc) * 16 + e] = ssum[e];
            red[(rr * CPR + cc) * 16 + 8 + e] = ssq[e];
        }
        __syncthreads();
        float* srow = a.stats + (size_t)(mt % a.stats_rows) * 2 * a.stats_c + a.stats_off;
        for (int col = t; col < 2 * BN; col += NT) {
            const int which = col / BN, ch = col - which * BN;
            if (n0 + ch < a.Cout) {
                float acc1 = 0.f;
                for (int r = 0; r < RPP; ++r) acc1 += red[(r * CPR + (ch >> 3)) * 16 + which * 8 + (ch & 7)];
                unsafeAtomicAdd(srow + which * a.stats_c + n0 + ch, acc1);
            }
        }
    }
#ifdef Y5_STAMPS
    if (lane == 0 && blockIdx.x * NW + wave < 65536) {
        unsigned long long* o = y5_stamp_buf + (size_t)(blockIdx.x * NW + wave) * 8;
        o[0] = st_wait; o[1] = st_issue; o[2] = st_mma; o[3] = st_loop_end - st_begin;
        o[4] = __builtin_amdgcn_s_memtime() - st_loop_end; o[5] = st_begin; o[6] = nk;
    }
#endif
}

#ifdef Y5_STAMPS
}  // namespace
extern "C" int y5_debug_stamps(unsigned long long* dst, int n_waves) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(y5_stamp_buf), (size_t)n_waves * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
namespace {
#endif

template <int BM, int BN, int WM, int WN, int S, int BK, typename ET = f16>
int launch_tile(const ConvArgs& a0, int ksize, bool direct, hipStream_t s) {
    ConvArgs a = a0;
    const long long tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.Cout + BN - 1) / BN;
    a.nk = (a.K + BK - 1) / BK;  // the weight rows are zero-padded to a multiple of 64 >= K
    const long long total = tiles_m * a.tiles_n;
    if (total <= 0 || total > 0x7fffffffLL) return y5_fail(Y5_EINVAL, "conv: tile count %lld out of range", total);
    a.tiles_total = (int)total;
    typedef IgemmCfg<BM, BN, WM, WN, S, BK> Cfg;
    const dim3 grid((unsigned)total), block(Cfg::NT);
    static std::atomic<unsigned long long> attr_set{0};  // one bit per device: the attribute is per function and device
    const unsigned long long attr_set_bit = 1ull << y5_current_device();
    if (!(attr_set.load() & attr_set_bit)) {
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<BM, BN, WM, WN, 3, S, BK, false, false, false, ET>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<BM, BN, WM, WN, 3, S, BK, false, true, false, ET>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<BM, BN, WM, WN, 1, S, BK, true, false, false, ET>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<BM, BN, WM, WN, 1, S, BK, false, false, false, ET>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
        attr_set.fetch_or(attr_set_bit);
    }
    // uniform-tap addressing whenever restarting the K-steps at every tap costs no extra step over walking K = 9 Cin
    // contiguously (always when Cin is a multiple of BK)
    const int nk_ut = 9 * ((a.Cin + BK - 1) / BK);
    const bool ut = ksize == 3 && nk_ut <= a.nk && !getenv("Y5_NO_UT");
    if (ut) a.nk = nk_ut;
    if (ut)
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, 3, S, BK, false, true, false, ET>), grid, block, Cfg::SMEM, s, a);
    else if (ksize == 3)
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, 3, S, BK, false, false, false, ET>), grid, block, Cfg::SMEM, s, a);
    else if (direct)
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, 1, S, BK, true, false, false, ET>), grid, block, Cfg::SMEM, s, a);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, 1, S, BK, false, false, false, ET>), grid, block, Cfg::SMEM, s, a);
    Y5_LAUNCH_CHECK("conv_igemm_kernel");
    return Y5_OK;
}

template <int BM, int BN, int WM, int WN, int S, int BK>
int launch_cat2(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    const long long tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.Cout + BN - 1) / BN;
    a.nk = (a.K + BK - 1) / BK;
    const long long total = tiles_m * a.tiles_n;
    if (total <= 0 || total > 0x7fffffffLL) return y5_fail(Y5_EINVAL, "conv_cat2: tile count %lld out of range", total);
    a.tiles_total = (int)total;
    typedef IgemmCfg<BM, BN, WM, WN, S, BK> Cfg;
    static std::atomic<unsigned long long> attr_set{0};
    const unsigned long long attr_set_bit = 1ull << y5_current_device();
    if (!(attr_set.load() & attr_set_bit)) {
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_igemm_kernel<BM, BN, WM, WN, 1, S, BK, true, false, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
        attr_set.fetch_or(attr_set_bit);
    }
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, 1, S, BK, true, false, true>), dim3((unsigned)total), dim3(Cfg::NT), Cfg::SMEM, s, a);
    Y5_LAUNCH_CHECK("conv_igemm_kernel<cat2>");
    return Y5_OK;
}

struct Cat2Variant {
    int bn;
    int (*launch)(const ConvArgs&, hipStream_t);
};
#define Y5_C2(BM, BN, WM, WN, S) {BN, &launch_cat2<BM, BN, WM, WN, S, 64>}
const Cat2Variant kCat2[] = {
    Y5_C2(128, 128, 2, 2, 2), Y5_C2(128, 128, 2, 4, 3), Y5_C2(256, 128, 4, 2, 2), Y5_C2(256, 128, 4, 2, 3),
    Y5_C2(128, 64, 2, 2, 3),  Y5_C2(64, 128, 2, 2, 3),  Y5_C2(256, 256, 2, 4, 2), Y5_C2(256, 64, 4, 2, 3),
    Y5_C2(128, 128, 2, 4, 2), Y5_C2(128, 128, 2, 2, 3), Y5_C2(64, 64, 2, 2, 3),   Y5_C2(128, 64, 4, 2, 3),
    Y5_C2(64, 128, 2, 4, 3),  Y5_C2(256, 64, 4, 2, 2),
};
#undef Y5_C2
constexpr int kNumCat2 = (int)(sizeof(kCat2) / sizeof(kCat2[0]));

// Every compiled tile configuration. Index + 1 is the public variant id (0 = choose by rule); ids past the end
// of this table select the other kernel families in turn (conv_halo.hip, conv_ws1x1.hip, conv3x3_ws.hip, conv_lw.hip,
// conv_lin3.hip).
struct Variant {
    int bm, bn, bk, s;
    int (*launch)(const ConvArgs&, int, bool, hipStream_t);
};
#define Y5_V(BM, BN, WM, WN, S, BK) {BM, BN, BK, S, &launch_tile<BM, BN, WM, WN, S, BK>}
const Variant kVariants[] = {
    Y5_V(128, 128, 2, 2, 2, 64), Y5_V(128, 128, 2, 2, 3, 32), Y5_V(128, 128, 2, 2, 4, 32),
    Y5_V(128, 64, 2, 2, 2, 64),  Y5_V(128, 64, 2, 2, 3, 64),  Y5_V(128, 64, 2, 2, 3, 32),  Y5_V(128, 64, 2, 2, 4, 32),
    Y5_V(64, 128, 2, 2, 2, 64),  Y5_V(64, 128, 2, 2, 3, 64),  Y5_V(64, 128, 2, 2, 3, 32),  Y5_V(64, 128, 2, 2, 4, 32),
    Y5_V(64, 64, 2, 2, 2, 64),   Y5_V(64, 64, 2, 2, 3, 64),   Y5_V(64, 64, 2, 2, 4, 64),   Y5_V(64, 64, 2, 2, 3, 32),
    Y5_V(64, 64, 2, 2, 4, 32),
    Y5_V(128, 32, 4, 1, 2, 64),  Y5_V(128, 32, 4, 1, 3, 64),  Y5_V(128, 32, 4, 1, 3, 32),  Y5_V(128, 32, 4, 1, 4, 32),
    // eight waves: the same LDS footprint hides twice the DMA latency
    Y5_V(128, 128, 2, 4, 2, 64), Y5_V(128, 128, 2, 4, 3, 64), Y5_V(128, 128, 2, 4, 3, 32), Y5_V(128, 128, 2, 4, 4, 32),
    Y5_V(256, 128, 4, 2, 2, 64), Y5_V(256, 128, 4, 2, 3, 32), Y5_V(256, 128, 4, 2, 2, 32),
    Y5_V(128, 64, 4, 2, 2, 64),  Y5_V(128, 64, 4, 2, 3, 64),  Y5_V(64, 128, 2, 4, 2, 64),  Y5_V(64, 128, 2, 4, 3, 64),
    // tall tiles for narrow layers: the weight tile is a smaller share of what every K-step stages
    Y5_V(256, 64, 4, 2, 2, 64),  Y5_V(256, 64, 4, 2, 3, 32),  Y5_V(256, 32, 8, 1, 2, 64),  Y5_V(256, 32, 8, 1, 3, 32),
    // 256 x 256: 128 MFMA FLOP per staged byte, for layers with enough pixels and channels to fill the chip with such tiles
    Y5_V(256, 256, 2, 4, 3, 32), Y5_V(256, 256, 2, 4, 2, 64), Y5_V(256, 256, 4, 2, 2, 64),
    // deeper rings on the big tiles: two K-steps (about 96 KB per CU) in flight cover the DMA round trip
    Y5_V(256, 128, 4, 2, 3, 64), Y5_V(256, 128, 4, 2, 4, 32), Y5_V(256, 256, 2, 4, 4, 32), Y5_V(256, 64, 4, 2, 3, 64),
};
#undef Y5_V
constexpr int kNumIgemm = (int)(sizeof(kVariants) / sizeof(kVariants[0]));
// The tile configurations compiled for bfloat16 tensors: every configuration the rule-based choice below can name, plus
// two wide ones for the layers with many channels.
#define Y5_VB(BM, BN, WM, WN, S, BK) {BM, BN, BK, S, &launch_tile<BM, BN, WM, WN, S, BK, bf16>}
const Variant kVariantsBf[] = {
    Y5_VB(128, 128, 2, 2, 2, 64), Y5_VB(128, 64, 2, 2, 3, 64),  Y5_VB(64, 128, 2, 2, 3, 64),  Y5_VB(64, 64, 2, 2, 3, 64),
    Y5_VB(128, 32, 4, 1, 3, 64),  Y5_VB(128, 128, 2, 2, 3, 32), Y5_VB(128, 64, 2, 2, 4, 32),  Y5_VB(64, 128, 2, 2, 4, 32),
    Y5_VB(64, 64, 2, 2, 4, 32),   Y5_VB(128, 32, 4, 1, 4, 32),  Y5_VB(256, 128, 4, 2, 2, 64), Y5_VB(128, 128, 2, 4, 3, 64),
};
#undef Y5_VB
constexpr int kNumBf = (int)(sizeof(kVariantsBf) / sizeof(kVariantsBf[0]));
constexpr int kNumHalo = 3;  // BN = 128, 64, 32
constexpr int kNumWs = 3;    // weight-stationary 1x1 kernel, BN = 128, 64, 32
constexpr int kNumWs3 = 2;   // weight-stationary 3x3 kernel for 32 / 64 input channels, BN = 64, 32

int find_variant(int bm, int bn, int bk, int s, const Variant* table = kVariants, int count = kNumIgemm) {
    for (int i = 0; i < count; ++i)
        if (table[i].bm == bm && table[i].bn == bn && table[i].bk == bk && table[i].s == s) return i;
    return 0;
}

}  // namespace

int y5_launch_conv(const y5_conv_desc& d, hipStream_t s, const Y5ConvStats* st) {
    Y5_REQUIRE(d.x && d.w && d.y, "conv: null tensor pointer");
    Y5_REQUIRE(d.ksize == 1 || d.ksize == 3, "conv: ksize %d not in {1,3}", d.ksize);
    Y5_REQUIRE(d.stride == 1 || d.stride == 2, "conv: stride %d not in {1,2}", d.stride);
    Y5_REQUIRE(d.pad == d.ksize / 2, "conv: pad %d != ksize/2", d.pad);
    Y5_REQUIRE(d.batch > 0 && d.h > 0 && d.w_in > 0 && d.ho > 0 && d.wo > 0, "conv: empty tensor");
    Y5_REQUIRE((long long)d.batch * d.ho * d.wo < 0x7fffffffLL && (long long)d.h * d.w_in < 0x7fffffffLL, "conv: more than 2^31 pixels");
    Y5_REQUIRE(d.ho == (d.h + 2 * d.pad - d.ksize) / d.stride + 1 && d.wo == (d.w_in + 2 * d.pad - d.ksize) / d.stride + 1,
               "conv: output %dx%d inconsistent with input %dx%d k%d s%d", d.ho, d.wo, d.h, d.w_in, d.ksize, d.stride);
    Y5_REQUIRE(d.cin > 0 && d.cin % 8 == 0 && d.x_ld % 8 == 0 && d.x_ld >= d.cin, "conv: cin %d / x_ld %d must be multiples of 8", d.cin, d.x_ld);
    Y5_REQUIRE(d.cout > 0 && d.cout % 8 == 0 && d.y_ld % 8 == 0 && d.y_ld >= d.cout, "conv: cout %d / y_ld %d must be multiples of 8", d.cout, d.y_ld);
    Y5_REQUIRE(d.kpad % 64 == 0 && d.kpad >= d.ksize * d.ksize * d.cin, "conv: kpad %d too small or not a multiple of 64", d.kpad);
    Y5_REQUIRE(d.w_rows % 128 == 0 && d.w_rows >= d.cout, "conv: w_rows %d must be a multiple of 128 >= cout", d.w_rows);
    Y5_REQUIRE(!d.scale || d.shift, "conv: scale without shift");
    Y5_REQUIRE(!d.res || (d.res_ld % 8 == 0 && d.res_ld >= d.cout), "conv: bad residual pitch %d", d.res_ld);
    Y5_REQUIRE(y5_aligned16(d.x) && y5_aligned16(d.w) && y5_aligned16(d.y) && y5_aligned16(d.res), "conv: pointers must be 16-byte aligned");
    Y5_REQUIRE(y5_aligned16(d.bias) && y5_aligned16(d.scale) && y5_aligned16(d.shift), "conv: per-channel vectors must be 16-byte aligned");
    Y5_REQUIRE(d.act == Y5_ACT_NONE || d.act == Y5_ACT_LEAKY, "conv: unknown activation %d", d.act);
    Y5_REQUIRE(d.act == Y5_ACT_NONE || (d.slope > 0.f && d.slope <= 1.f), "conv: LeakyReLU slope %g outside (0, 1]", (double)d.slope);

    Y5_REQUIRE(d.dtype == Y5_F16 || d.dtype == Y5_BF16, "conv: dtype %d not in {Y5_F16, Y5_BF16}", d.dtype);
    const bool bf = d.dtype == Y5_BF16;
    if (bf) {
        Y5_REQUIRE(!st, "conv: statistics are accumulated for fp16 tensors only");
        if (d.variant > kNumBf + y5_conv_ph_count()) return y5_fail(Y5_EINVAL, "conv: variant %d is not compiled for bf16 (1..%d are)", d.variant, kNumBf + y5_conv_ph_count());
        if (d.variant > kNumBf) {  // the phased kernels (conv_ph.hip) follow the im2col tile configurations
            int rc = Y5_OK;
            if (y5_try_conv_ph(d, s, d.variant - kNumBf, &rc)) return rc;
            return y5_fail(Y5_EINVAL, "conv: the phased variant %d does not apply to this shape", d.variant);
        }
    }
    Y5_REQUIRE(d.variant >= 0 && d.variant <= y5_conv_variant_count(), "conv: variant %d out of range 0..%d", d.variant, y5_conv_variant_count());
    if (d.variant > y5_conv_variant_count() - y5_conv_ph_count()) {  // the phased kernels (conv_ph.hip) come last
        Y5_REQUIRE(!st, "conv: variant %d does not accumulate statistics", d.variant);
        int rc = Y5_OK;
        if (y5_try_conv_ph(d, s, d.variant - (y5_conv_variant_count() - y5_conv_ph_count()), &rc)) return rc;
        return y5_fail(Y5_EINVAL, "conv: the phased variant %d does not apply to this shape", d.variant);
    }
    if (st) {
        Y5_REQUIRE(st->partials && st->rows > 0 && st->c > 0 && st->off >= 0 && st->off + d.cout <= st->c, "conv: statistics columns %d..+%d outside 0..%d (rows %d)", st->off, d.cout, st->c, st->rows);
        Y5_REQUIRE((reinterpret_cast<uintptr_t>(st->partials) & 3) == 0, "conv: misaligned statistics pointer");
        if (d.variant > kNumIgemm) return y5_fail(Y5_EINVAL, "conv: variant %d does not accumulate statistics (variants 1..%d do)", d.variant, kNumIgemm);
    }
    // the persistent kernels for single shape classes come last: 1x1 (conv_p1x1.hip), 3x3 stride 2 from 64 and from 32 channels
    const int kBase = kNumIgemm + kNumHalo + kNumWs + kNumWs3 + y5_conv_lw_count() + y5_conv_lin_count();
    if (d.variant > kBase + y5_conv_p1x1_count() + y5_conv_s2c64_count()) {
        int rc = Y5_OK;
        if (y5_try_conv_s2c32(d, s, d.variant - kBase - y5_conv_p1x1_count() - y5_conv_s2c64_count(), &rc)) return rc;
        return y5_fail(Y5_EINVAL, "conv: the persistent stride-2 variant %d takes 3x3 stride 2, 32 -> 64 channels, output height %% 8 == 0 and width %% 16 == 0", d.variant);
    }
    if (d.variant > kBase + y5_conv_p1x1_count()) {
        int rc = Y5_OK;
        if (y5_try_conv_s2c64(d, s, &rc)) return rc;
        return y5_fail(Y5_EINVAL, "conv: the persistent stride-2 variant %d takes 3x3 stride 2, 64 -> 128 channels, output height %% 4 == 0 and width %% 16 == 0", d.variant);
    }
    if (d.variant > kBase) {
        int rc = Y5_OK;
        if (y5_try_conv_p1x1(d, s, d.variant - kBase, &rc)) return rc;
        return y5_fail(Y5_EINVAL, "conv: the persistent 1x1 variant %d takes 1x1 stride 1 with 128 or 256 input and output channels, no residual", d.variant);
    }
    if (d.variant > kNumIgemm + kNumHalo + kNumWs + kNumWs3 + y5_conv_lw_count()) {
        int rc = Y5_OK;
        if (y5_try_conv_lin(d, s, d.variant - kNumIgemm - kNumHalo - kNumWs - kNumWs3 - y5_conv_lw_count(), &rc)) return rc;
        return y5_fail(Y5_EINVAL, "conv: the linear-halo variant %d does not apply to this shape", d.variant);
    }
    if (d.variant > kNumIgemm + kNumHalo + kNumWs + kNumWs3) {
        int rc = Y5_OK;
        if (y5_try_conv_lw(d, s, d.variant - kNumIgemm - kNumHalo - kNumWs - kNumWs3, &rc)) return rc;
        return y5_fail(Y5_EINVAL, "conv: the loader-wave variant %d does not apply to this shape", d.variant);
    }
    if (d.variant > kNumIgemm + kNumHalo + kNumWs) {
        int rc = Y5_OK;
        if (y5_try_conv3x3_ws(d, s, d.variant - kNumIgemm - kNumHalo - kNumWs, &rc)) return rc;
        return y5_fail(Y5_EINVAL, "conv: the weight-stationary 3x3 variant %d does not apply to this shape", d.variant);
    }
    if (d.variant > kNumIgemm + kNumHalo) {
        int rc = Y5_OK;
        if (y5_try_conv_ws1x1(d, s, d.variant - kNumIgemm - kNumHalo, &rc)) return rc;
        return y5_fail(Y5_EINVAL, "conv: the weight-stationary variant %d does not apply to this shape", d.variant);
    }
    if (!bf && (d.variant > kNumIgemm || (d.variant == 0 && !st && !getenv("Y5_NO_HALO")))) {
        int rc = Y5_OK;
        const int bn = d.variant == 0 ? 0 : (128 >> (d.variant - kNumIgemm - 1));
        if (y5_try_conv_halo(d, s, bn, &rc)) return rc;
        if (d.variant != 0) return y5_fail(Y5_EINVAL, "conv: the halo-tile variant %d does not apply to this shape", d.variant);
    }
    // The im2col kernel addresses x and w by 32-bit byte offsets from their first element.  A batch whose input spans
    // 2 GiB or more is convolved in slices of whole images.
    Y5_REQUIRE((long long)d.w_rows * d.kpad * 2 < 0x7fffffffLL, "conv: packed weights of 2 GiB or more");
    const long long img_span = (long long)d.h * d.w_in * d.x_ld * 2;
    const long long x_span = (((long long)d.batch * d.h * d.w_in - 1) * d.x_ld + d.cin) * 2;
    if (x_span >= 0x7fffffffLL) {
        Y5_REQUIRE(img_span < 0x7fffffffLL, "conv: one input image of 2 GiB or more");
        const int per = (int)(0x7fffffffLL / img_span);
        for (int b0 = 0; b0 < d.batch; b0 += per) {
            y5_conv_desc part = d;
            part.batch = d.batch - b0 < per ? d.batch - b0 : per;
            part.x = static_cast<const f16*>(d.x) + (size_t)b0 * d.h * d.w_in * d.x_ld;
            part.y = static_cast<f16*>(d.y) + (size_t)b0 * d.ho * d.wo * d.y_ld;
            if (d.res) part.res = static_cast<const f16*>(d.res) + (size_t)b0 * d.ho * d.wo * d.res_ld;
            const int rc = y5_launch_conv(part, s, st);
            if (rc != Y5_OK) return rc;
        }
        return Y5_OK;
    }
    ConvArgs a{};
    a.x = static_cast<const f16*>(d.x);
    a.w = static_cast<const f16*>(d.w);
    a.bias = d.bias;
    a.scale = d.scale;
    a.shift = d.shift;
    a.res = static_cast<const f16*>(d.res);
    a.y = static_cast<f16*>(d.y);
    a.M = (long long)d.batch * d.ho * d.wo;
    a.H = d.h; a.W = d.w_in; a.Cin = d.cin; a.x_ld = d.x_ld;
    a.Ho = d.ho; a.Wo = d.wo; a.Cout = d.cout; a.y_ld = d.y_ld; a.res_ld = d.res_ld;
    a.stride = d.stride; a.pad = d.pad; a.K = d.ksize * d.ksize * d.cin;
    a.w_ld = d.kpad; a.nk = 0;
    a.tiles_n = 0; a.tiles_total = 0;
    a.slope = d.act == Y5_ACT_LEAKY ? d.slope : 1.f;
    a.div_hw = make_fastdiv((unsigned)(d.ho * d.wo));
    a.div_wo = make_fastdiv((unsigned)d.wo);
    a.div_cin = make_fastdiv((unsigned)d.cin);
    const bool direct = d.ksize == 1 && d.stride == 1;
    a.x_bytes = (int)x_span;
    a.w_bytes = (int)((long long)d.w_rows * d.kpad * 2);
    if (st) {
        a.stats = st->partials;
        a.stats_rows = st->rows; a.stats_c = st->c; a.stats_off = st->off;
    }

    const Variant* table = bf ? kVariantsBf : kVariants;
    const int count = bf ? kNumBf : kNumIgemm;
    if (d.variant > 0) {
        // a tile reads BN whole weight rows: the packed weights (rows padded to a multiple of 128) must hold them
        const int bn = table[d.variant - 1].bn;
        if ((long long)((d.cout + bn - 1) / bn) * bn > d.w_rows)
            return y5_fail(Y5_EINVAL, "conv: variant %d reads %d-row weight tiles, w_rows %d is too small for cout %d", d.variant, bn, d.w_rows, d.cout);
        return table[d.variant - 1].launch(a, d.ksize, direct, s);
    }

    // Rule-based choice. Channel width: the widest of {128, 64, 32} that wastes the fewest padded channels.
    // Pixel height: 128 rows unless that leaves the 256 CUs with fewer than ~4 workgroups each, in which
    // case 64-row tiles double the number of independent DMA streams.  K-step: 64 (whole 128-byte lines per
    // row) unless K is so short that rounding it up to 64 would mostly move padding.
    auto waste = [&](int bn) { return ((d.cout + bn - 1) / bn) * bn - d.cout; };
    int bn = 128;
    if (waste(64) < waste(bn)) bn = 64;
    if (waste(32) < waste(bn) && d.cout <= 32) bn = 32;
    const long long blocks128 = ((a.M + 127) / 128) * ((d.cout + bn - 1) / bn);
    const bool small = blocks128 < 1024 && bn != 32;
    const int tail = a.K % 64;
    const bool k32 = a.K < 256 && tail != 0 && tail <= 32;
    const int bm = small ? 64 : 128;
    int v;
    if (k32) v = find_variant(bm, bn, 32, (bm == 128 && bn == 128) ? 3 : 4, table, count);
    else v = find_variant(bm, bn, 64, (bm == 128 && bn == 128) ? 2 : 3, table, count);
    return table[v].launch(a, d.ksize, direct, s);
}

int y5_conv_variant_count() { return kNumIgemm + kNumHalo + kNumWs + kNumWs3 + y5_conv_lw_count() + y5_conv_lin_count() + y5_conv_p1x1_count() + y5_conv_s2c64_count() + y5_conv_s2c32_count() + y5_conv_ph_count(); }

int y5_conv_cat2_variant_count() { return kNumCat2 + 1; }  // the tile configurations, then the persistent kernel (conv_p1x1.hip)
int y5_conv_bf16_variant_count() { return kNumBf + y5_conv_ph_count(); }

// d: the convolution with x = the half-resolution tensor (batch, ho/2, wo/2, cin), h = ho, w_in = wo the OUTPUT
// resolution; x2 (batch, ho, wo, cin2).  Weight rows hold K = cin + cin2 values, x's channels first.
int y5_launch_conv_cat2(const y5_conv_desc& d, const void* x2, int x2_ld, int cin2, hipStream_t s) {
    Y5_REQUIRE(d.dtype == Y5_F16, "conv_cat2: fp16 only (dtype %d)", d.dtype);
    Y5_REQUIRE(d.x && x2 && d.w && d.y, "conv_cat2: null tensor pointer");
    Y5_REQUIRE(d.ksize == 1 && d.stride == 1 && d.pad == 0, "conv_cat2: only 1x1 stride-1 convolutions");
    Y5_REQUIRE(d.batch > 0 && d.ho > 0 && d.wo > 0 && d.h == d.ho && d.w_in == d.wo, "conv_cat2: h, w_in must equal the output size %dx%d", d.ho, d.wo);
    Y5_REQUIRE(d.ho % 2 == 0 && d.wo % 2 == 0, "conv_cat2: output %dx%d must be even (2x upsampled source)", d.ho, d.wo);
    Y5_REQUIRE((long long)d.batch * d.ho * d.wo < 0x7fffffffLL, "conv_cat2: more than 2^31 pixels");
    Y5_REQUIRE(d.cin > 0 && d.cin % 64 == 0 && d.x_ld % 8 == 0 && d.x_ld >= d.cin, "conv_cat2: cin %d must be a multiple of 64, x_ld %d of 8", d.cin, d.x_ld);
    Y5_REQUIRE(cin2 > 0 && cin2 % 8 == 0 && x2_ld % 8 == 0 && x2_ld >= cin2, "conv_cat2: cin2 %d / x2_ld %d must be multiples of 8", cin2, x2_ld);
    Y5_REQUIRE(d.cout > 0 && d.cout % 8 == 0 && d.y_ld % 8 == 0 && d.y_ld >= d.cout, "conv_cat2: cout %d / y_ld %d must be multiples of 8", d.cout, d.y_ld);
    Y5_REQUIRE(d.kpad % 64 == 0 && d.kpad >= d.cin + cin2, "conv_cat2: kpad %d too small or not a multiple of 64", d.kpad);
    Y5_REQUIRE(d.w_rows % 128 == 0 && d.w_rows >= d.cout, "conv_cat2: w_rows %d must be a multiple of 128 >= cout", d.w_rows);
    Y5_REQUIRE(!d.scale || d.shift, "conv_cat2: scale without shift");
    Y5_REQUIRE(!d.res || (d.res_ld % 8 == 0 && d.res_ld >= d.cout), "conv_cat2: bad residual pitch %d", d.res_ld);
    Y5_REQUIRE(y5_aligned16(d.x) && y5_aligned16(x2) && y5_aligned16(d.w) && y5_aligned16(d.y) && y5_aligned16(d.res), "conv_cat2: pointers must be 16-byte aligned");
    Y5_REQUIRE(y5_aligned16(d.bias) && y5_aligned16(d.scale) && y5_aligned16(d.shift), "conv_cat2: per-channel vectors must be 16-byte aligned");
    Y5_REQUIRE(d.act == Y5_ACT_NONE || d.act == Y5_ACT_LEAKY, "conv_cat2: unknown activation %d", d.act);
    Y5_REQUIRE(d.act == Y5_ACT_NONE || (d.slope > 0.f && d.slope <= 1.f), "conv_cat2: LeakyReLU slope %g outside (0, 1]", (double)d.slope);
    Y5_REQUIRE(d.variant >= 0 && d.variant <= kNumCat2 + 1, "conv_cat2: variant %d out of range 0..%d", d.variant, kNumCat2 + 1);
    const long long x_span = (((long long)d.batch * (d.ho / 2) * (d.wo / 2) - 1) * d.x_ld + d.cin) * 2;
    const long long x2_span = (((long long)d.batch * d.ho * d.wo - 1) * x2_ld + cin2) * 2;
    Y5_REQUIRE(x_span < 0x7fffffffLL && x2_span < 0x7fffffffLL && (long long)d.w_rows * d.kpad * 2 < 0x7fffffffLL, "conv_cat2: a tensor of 2 GiB or more");
    if (d.variant == kNumCat2 + 1) {
        int rc = Y5_OK;
        if (y5_try_conv_p1x1_cat2(d, x2, x2_ld, cin2, s, &rc)) return rc;
        return y5_fail(Y5_EINVAL, "conv_cat2: the persistent variant %d takes 128 + 128 -> 128 or 256 + 256 -> 256 channels, no residual", d.variant);
    }
    ConvArgs a{};
    a.x = static_cast<const f16*>(d.x);
    a.x2 = static_cast<const f16*>(x2);
    a.w = static_cast<const f16*>(d.w);
    a.bias = d.bias; a.scale = d.scale; a.shift = d.shift;
    a.res = static_cast<const f16*>(d.res);
    a.y = static_cast<f16*>(d.y);
    a.M = (long long)d.batch * d.ho * d.wo;
    a.H = d.ho; a.W = d.wo; a.Cin = d.cin; a.x_ld = d.x_ld; a.x2_ld = x2_ld;
    a.Ho = d.ho; a.Wo = d.wo; a.Cout = d.cout; a.y_ld = d.y_ld; a.res_ld = d.res_ld;
    a.stride = 1; a.pad = 0; a.K = d.cin + cin2;
    a.w_ld = d.kpad;
    a.slope = d.act == Y5_ACT_LEAKY ? d.slope : 1.f;
    a.div_hw = make_fastdiv((unsigned)(d.ho * d.wo));
    a.div_wo = make_fastdiv((unsigned)d.wo);
    a.div_cin = make_fastdiv((unsigned)d.cin);
    a.x_bytes = (int)x_span; a.x2_bytes = (int)x2_span;
    a.w_bytes = (int)((long long)d.w_rows * d.kpad * 2);
    const Cat2Variant& v = kCat2[d.variant > 0 ? d.variant - 1 : 0];
    if ((long long)((d.cout + v.bn - 1) / v.bn) * v.bn > d.w_rows)
        return y5_fail(Y5_EINVAL, "conv_cat2: variant %d reads %d-row weight tiles, w_rows %d is too small for cout %d", d.variant, v.bn, d.w_rows, d.cout);
    return v.launch(a, s);
}
