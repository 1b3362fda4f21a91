// 3x3 stride-1 convolution with the pixel operand resident in LDS ("linear halo") for gfx950.
//
// The batch is one list of pixels m = (b*H + y)*W + x, NHWC.  Output pixel m reads, for tap (r, q), input pixel
// m + (r-1)*W + (q-1) - whenever that pixel lies in the same image.  A workgroup owns BM consecutive output pixels and BN
// output channels; the input pixels [m0 - W - 1, m0 + BM + W + 1) it can touch are ONE contiguous range of memory.  The
// loader waves copy that range into LDS once (plain contiguous 1 KiB DMAs, no im2col gather), and all nine taps of all
// K-steps read it at shifted offsets; a tap that leaves the image (left / right / top / bottom border, or a tile that
// runs over an image boundary) is redirected, per lane, to a pixel of zeros.  Only the weights stream through a ring:
// per K-step BN rows x BK columns, a third of the bytes an im2col K-step stages, so the loader waves' DMA issue rate -
// the limit of the im2col kernels (conv_lw.hip) on these layers - is out of the way.
//
// LDS rows are padded by one 16-byte slot (pixels: Cin*2 + 16 bytes, weight rows: BK*2 + 16): with an odd number of
// slots per row, 16 consecutive rows at the same column fall on 16 different 16-byte bank groups, so ds_read_b128
// fragments are conflict-free without an XOR swizzle and every k-sub-step is an immediate offset from one base address.
//
// The accumulators go to memory straight from registers: weight rows are staged in the order (16*h + 4*g + j) for MFMA
// row 8*g + 4*h + j, which makes the 16 values a lane holds of a 32x32 tile 16 CONSECUTIVE channels of one pixel: post-ops
// in registers, two 16-byte stores, no LDS staging and no barrier behind the K loop.
#include "y5_common.h"

#include <cstdlib>

namespace {

#include "conv_common.h"

template <int BM, int BN, int WM, int WN, int NL, int S, int BK>
struct LinCfg {
    static constexpr int NWC = WM * WN;
    static constexpr int NT = 64 * (NWC + NL);
    static constexpr int WROW = BK * 2 + 16;                       // bytes per staged weight row
    static constexpr int STAGE_PIECES = (BN * WROW + 1023) / 1024;  // 1 KiB DMA pieces per K-step
    static constexpr int STAGE_ALLOC = STAGE_PIECES * 1024;
    static constexpr int RING = S * STAGE_ALLOC;
    static constexpr int PRM0 = RING;                               // [bias | scale | shift][BN] floats
    static constexpr int ZERO0 = PRM0 + 3 * BN * 4;                 // one pixel of zeros (up to 2 KiB + 16)
    static constexpr int ZERO_BYTES = 2048 + 64;
    static constexpr int HALO0 = (ZERO0 + ZERO_BYTES + 1023) / 1024 * 1024;
};

#ifdef Y5_STAMPS
// Diagnostic build only (make dbg): per consumer wave {start, first barrier passed, K loop end, end, cycles in barriers}
__device__ unsigned long long y5_lin_stamp_buf[8 * 65536];
#endif

__device__ __forceinline__ void lgkm_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// MODE bit 0: the epilogue stores straight from the accumulator registers (two 16-byte stores per MFMA tile and lane, no
// barrier behind the K loop) instead of transposing the tile through LDS; bit 1: no stagger between the consumer halves.
// Which form is faster depends on the layer (narrow tiles with short K loops prefer the direct form), so both are compiled
// and the caller's autotuner decides.
template <int BM, int BN, int WM, int WN, int NL, int S, int BK, int MODE>
__global__ __launch_bounds__(64 * (WM * WN + NL)) void conv_lin3_kernel(const ConvArgs a) {
    typedef LinCfg<BM, BN, WM, WN, NL, S, BK> Cfg;
    constexpr int NWC = Cfg::NWC, WROW = Cfg::WROW, WSLOTS = WROW / 16;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int RM = WTM / 32, RN = WTN / 32;
    static_assert(RM >= 1 && RN >= 1 && WTM % 32 == 0 && WTN % 32 == 0, "wave tiles are whole 32x32 MFMA tiles");
    static_assert(BK == 32 || BK == 64, "K-step is 32 or 64 channels of one tap");
    static_assert(S == 3 && NWC % 2 == 0, "the staggered slot schedule needs a ring of three stages and two halves of consumer waves");
    constexpr int LP = (Cfg::STAGE_PIECES + NL - 1) / NL;  // most pieces a loader issues per K-step
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tile = xcd_remap(blockIdx.x, a.tiles_total);
    const int mt = tile / a.tiles_n, nt = tile - mt * a.tiles_n;
    const int m0 = mt * BM, n0 = nt * BN;
    const int nk = a.nk;
    const int pitch = a.lin_pitch;

    // per-channel parameters and the pixel of zeros; published by the first K-step barrier
    float* prm = reinterpret_cast<float*>(smem + Cfg::PRM0);
    epi_store(prm, epi_fetch(a, n0, t, BN), t, BN);
    if (t < Cfg::ZERO_BYTES / 16) reinterpret_cast<uint4*>(smem + Cfg::ZERO0)[t] = uint4{0u, 0u, 0u, 0u};

    if (wave >= NWC) {
        // ------------------------------------------------------------------ loader waves
        const int lw = wave - NWC;
        constexpr unsigned OOB = 0x80000000u;
        const BufRsrc xr = make_rsrc(a.x, a.x_bytes), wr = make_rsrc(a.w, a.w_bytes);
        // the halo: staged pixel p is input pixel m0 - W - 1 + p, slot `within` of it its channels [8*within, +8)
        {
            const long long g0 = (long long)m0 - a.W - 1;
            const unsigned slots = (unsigned)pitch >> 4, cin_slots = (unsigned)a.Cin >> 3;
            for (int pi = lw; pi < a.lin_pieces; pi += NL) {
                const unsigned lc = (unsigned)pi * 64u + (unsigned)lane;
                const unsigned p = fdiv(lc, a.div_slots);
                const unsigned within = lc - p * slots;
                const long long g = g0 + (long long)p;
                const bool ok = within < cin_slots && p < (unsigned)a.lin_hpx && g >= 0 && g < a.lin_min;
                const unsigned vo = ok ? (unsigned)g * (unsigned)(a.x_ld * 2) + within * 16u : OOB;
                buf_dma16(xr, (lds_ptr)(smem + Cfg::HALO0 + pi * 1024), vo, 0);
            }
        }
        // the weight ring: LDS row `row` of a stage holds output channel n0 + 32*(row/32) + perm(row%32)
        unsigned wbase[LP];
        int mine = 0;
#pragma unroll
        for (int i = 0; i < LP; ++i) {
            const int pj = lw + NL * i;
            const unsigned lc = (unsigned)pj * 64u + (unsigned)lane;
            const unsigned row = lc / WSLOTS, within = lc - row * WSLOTS;
            const unsigned ri = row & 31u;
            const unsigned ch = (row & ~31u) + 16u * ((ri >> 2) & 1u) + 4u * (ri >> 3) + (ri & 3u);
            const bool ok = pj < Cfg::STAGE_PIECES && row < (unsigned)BN && within < (unsigned)(BK / 8);
            wbase[i] = ok ? ((unsigned)n0 + ch) * (unsigned)(a.w_ld * 2) + within * 16u : OOB;
            if (pj < Cfg::STAGE_PIECES) ++mine;
        }
        mine = __builtin_amdgcn_readfirstlane(mine);
        int koff = 0;  // byte offset of the K-step in a weight row: (tap*Cin + c0) * 2, and consecutive steps are adjacent
        auto issue = [&](int stage) {
            unsigned char* base = smem + stage * Cfg::STAGE_ALLOC;
#pragma unroll
            for (int i = 0; i < LP; ++i) {
                const int pj = lw + NL * i;
                if (pj < Cfg::STAGE_PIECES) {
                    const unsigned vo = wbase[i] == OOB ? OOB : wbase[i] + (unsigned)koff;
                    buf_dma16(wr, (lds_ptr)(base + pj * 1024), vo, 0);
                }
            }
            koff += BK * 2;
        };
        // Slot schedule (one barrier per slot, 2*nk + 1 slots): the first half of the consumers multiplies stage kt in
        // slots 2kt and 2kt+1, the second half one slot later.  Stage kt has landed before the barrier of slot 2kt; its
        // buffer is free again after the barrier of slot 2kt+3, which is when stage kt+3 is issued into it.
        int issued = 0;
#pragma unroll
        for (int s = 0; s < S; ++s)
            if (s < nk) {
                issue(s);
                ++issued;
            }
        for (int sl = 0; sl <= 2 * nk; ++sl) {
            if (!(sl & 1) && (sl >> 1) < nk) {
#ifdef Y5_STAMPS
                if (a.stats_rows & 1) { wait_vm_and_barrier(0); continue; }
#endif
                wait_vm_and_barrier((issued - (sl >> 1) - 1) * mine);  // the halo is older than every stage
            } else {
#ifdef Y5_STAMPS
                if (!(a.stats_rows & 8) || sl == 2 * nk)
#endif
                lgkm_barrier();
                const int j = (sl + 3) >> 1;
#ifdef Y5_STAMPS
                if (a.stats_rows & 1) continue;  // diagnostic: no weight DMAs behind the prologue (stale stages are multiplied)
#endif
                if (sl >= 3 && j < nk) {
                    issue(j % S);
                    ++issued;
                }
            }
        }
        if constexpr (!(MODE & 1)) lgkm_barrier();  // the consumers' barrier in front of the epilogue (LDS is reused there)
        return;
    }

    // ---------------------------------------------------------------------- consumer waves
    const int wm = wave % WM, wn = wave / WM;
    const int lr = lane & 31, lh = lane >> 5;
    // per M-fragment: halo index of the pixel at tap (0, 0) and which taps stay inside its image
    int pl[RM];
    unsigned vmask[RM];
#pragma unroll
    for (int i = 0; i < RM; ++i) {
        const int ml = wm * WTM + i * 32 + lr;
        const long long m = (long long)m0 + ml;
        pl[i] = ml;
        vmask[i] = 0;
        if (m < a.M) {
            const unsigned mm = (unsigned)m;
            const unsigned b = fdiv(mm, a.div_hw);
            const unsigned rem = mm - b * (unsigned)(a.H * a.W);
            const int y = (int)fdiv(rem, a.div_wo), x = (int)rem - y * a.W;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int yy = y + tp / 3 - 1, xx = x + tp % 3 - 1;
                if ((unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W) vmask[i] |= 1u << tp;
            }
        }
    }
    f32x16 acc[RN][RM];
#pragma unroll
    for (int j = 0; j < RN; ++j)
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;

#ifdef Y5_STAMPS
    const unsigned long long st_begin = __builtin_amdgcn_s_memtime();
    unsigned long long st_first = 0, st_bar = 0;
#endif
    const int kc = a.Cin / BK;  // K-steps per tap
    int tap = 0, cc = 0, stage = 0;
    unsigned baddr[RM];
    const unsigned wrow0 = (unsigned)((wn * WTN + lr) * WROW + lh * 16);
    // The two halves of the consumer waves (waves w and w + NWC/2 share a SIMD) run half a K-step apart: at every barrier
    // one of the two has its fragments in registers already and keeps the SIMD's matrix pipe busy while the other
    // waits for the LDS reads of its new stage.
    const bool late = !(MODE & 2) && wave >= NWC / 2;
    constexpr int KS = BK / 16, HK = KS / 2;
    if (late) lgkm_barrier();  // slot 0: nothing to do yet
    for (int kt = 0; kt < nk; ++kt) {
        if (cc == 0) {
            const int r = (tap * 11) >> 5, q = tap - 3 * r;
#pragma unroll
            for (int i = 0; i < RM; ++i)
                baddr[i] = ((vmask[i] >> tap) & 1u) ? (unsigned)(Cfg::HALO0 + (pl[i] + r * a.W + q) * pitch + lh * 16)
                                                    : (unsigned)(Cfg::ZERO0 + lh * 16);
        }
        lgkm_barrier();  // the loaders saw stage kt (and the halo) land
#ifdef Y5_STAMPS
        if (kt == 0) st_first = __builtin_amdgcn_s_memtime();
#endif
        const unsigned char* A = smem + stage * Cfg::STAGE_ALLOC + wrow0;
        const unsigned coff = (unsigned)(cc * BK * 2);
        f16x8 xf[KS][RM], wf[KS][RN];
#ifdef Y5_STAMPS
        if (a.stats_rows & 2) {  // diagnostic: no fragment reads (registers hold arbitrary values)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int i = 0; i < RM; ++i) asm volatile("" : "=v"(xf[ks][i]));
#pragma unroll
                for (int j = 0; j < RN; ++j) asm volatile("" : "=v"(wf[ks][j]));
            }
        } else
#endif
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int i = 0; i < RM; ++i) xf[ks][i] = *reinterpret_cast<const f16x8*>(smem + baddr[i] + coff + ks * 32);
#pragma unroll
            for (int j = 0; j < RN; ++j) wf[ks][j] = *reinterpret_cast<const f16x8*>(A + j * 32 * WROW + ks * 32);
        }
#pragma unroll
        for (int ks = 0; ks < HK; ++ks)
#pragma unroll
            for (int j = 0; j < RN; ++j)
#pragma unroll
                for (int i = 0; i < RM; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks][j], xf[ks][i], acc[j][i], 0, 0, 0);
#ifdef Y5_STAMPS
        if (!(a.stats_rows & 8))  // diagnostic: one barrier per stage instead of two
#endif
        lgkm_barrier();  // second half of the stage: fragments are in registers
#pragma unroll
        for (int ks = HK; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < RN; ++j)
#pragma unroll
                for (int i = 0; i < RM; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks][j], xf[ks][i], acc[j][i], 0, 0, 0);
        if (++cc == kc) {
            cc = 0;
            ++tap;
        }
        if (++stage == S) stage = 0;
    }
    if (!late) lgkm_barrier();  // last slot: the late half finishes its last stage

#ifdef Y5_STAMPS
    const unsigned long long st_loop_end = __builtin_amdgcn_s_memtime();
#endif
    // ---- epilogue.  Lane (lr, lh) holds, of MFMA tile (j, i), channels [16*lh, 16*lh + 16) of the tile's 32 for pixel lr:
    // element e = 4*g + jj is channel 16*lh + e (weight rows were staged in that order).  Post-ops in registers, then the
    // wave's own WTM x WTN tile goes through a private LDS region (two 16-byte writes per MFMA tile) so that the global
    // stores are whole rows of WTN channels: 64 scattered 16-byte pieces per store instruction cost the texture unit
    // four times what the transposed form does.
    const bool has_bias = a.bias != nullptr, has_scale = a.scale != nullptr;
    if constexpr (MODE & 1) {
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const int nl = wn * WTN + j * 32 + 16 * lh;
            const int n = n0 + nl;
            if (n >= a.Cout) continue;
            const bool second = n + 8 < a.Cout;
#pragma unroll
            for (int i = 0; i < RM; ++i) {
                const long long m = (long long)m0 + wm * WTM + i * 32 + lr;
                if (m >= a.M) continue;
                f16x4 h[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]};
                    h[g] = post_op4_lds(v, nl + 4 * g, prm, BN, has_bias, has_scale, a.slope);
                }
                f16x8 lo = __builtin_shufflevector(h[0], h[1], 0, 1, 2, 3, 4, 5, 6, 7);
                f16x8 hi = __builtin_shufflevector(h[2], h[3], 0, 1, 2, 3, 4, 5, 6, 7);
                if (a.res) {
                    const f16* rp = a.res + (size_t)m * a.res_ld + n;
                    lo += *reinterpret_cast<const f16x8*>(rp);
                    if (second) hi += *reinterpret_cast<const f16x8*>(rp + 8);
                }
                f16* yp = a.y + (size_t)m * a.y_ld + n;
                *reinterpret_cast<f16x8*>(yp) = lo;
                if (second) *reinterpret_cast<f16x8*>(yp + 8) = hi;
            }
        }
    } else {
    lgkm_barrier();  // every consumer is done with the halo and the ring: their space is reused below
    constexpr int EPW = WTN * 2 + 16;
    unsigned char* ep = smem + Cfg::HALO0 + wave * (WTM * EPW);
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int nl = wn * WTN + j * 32 + 16 * lh;
#pragma unroll
        for (int i = 0; i < RM; ++i) {
            f16x4 h[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]};
                h[g] = post_op4_lds(v, nl + 4 * g, prm, BN, has_bias, has_scale, a.slope);
            }
            unsigned char* dst = ep + (i * 32 + lr) * EPW + (j * 32 + 16 * lh) * 2;
            *reinterpret_cast<f16x8*>(dst) = __builtin_shufflevector(h[0], h[1], 0, 1, 2, 3, 4, 5, 6, 7);
            *reinterpret_cast<f16x8*>(dst + 16) = __builtin_shufflevector(h[2], h[3], 0, 1, 2, 3, 4, 5, 6, 7);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own writes; no other wave touches this region
    constexpr int CPR = WTN / 8, PPI = 64 / CPR;
    static_assert(64 % CPR == 0 && WTM % PPI == 0, "a store instruction covers whole tile rows");
    const int c8 = lane % CPR, pr = lane / CPR;
    const int n = n0 + wn * WTN + c8 * 8;
#pragma unroll
    for (int p0 = 0; p0 < WTM; p0 += PPI) {
        const int px = p0 + pr;
        const long long m = (long long)m0 + wm * WTM + px;
        if (m < a.M && n < a.Cout) {
            f16x8 v = *reinterpret_cast<const f16x8*>(ep + px * EPW + c8 * 16);
            if (a.res) v += *reinterpret_cast<const f16x8*>(a.res + (size_t)m * a.res_ld + n);
            *reinterpret_cast<f16x8*>(a.y + (size_t)m * a.y_ld + n) = v;
        }
    }
    }
#ifdef Y5_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && blockIdx.x * NWC + wave < 65536) {
        unsigned long long* o = y5_lin_stamp_buf + (size_t)(blockIdx.x * NWC + wave) * 8;
        o[0] = st_begin; o[1] = st_first; o[2] = st_loop_end; o[3] = __builtin_amdgcn_s_memtime(); o[4] = st_bar; o[5] = (unsigned long long)nk;
    }
#endif
}

constexpr int kNotApplicable = 1;  // launch_lin: the tile configuration does not take the shape (no error recorded)

template <int BM, int BN, int WM, int WN, int NL, int S, int BK, int MODE>
int launch_lin(const ConvArgs& a0, hipStream_t s) {
    typedef LinCfg<BM, BN, WM, WN, NL, S, BK> Cfg;
    ConvArgs a = a0;
    if (a.Cin % BK) return kNotApplicable;
    a.lin_pitch = a.Cin * 2 + 16;
    if (a.lin_pitch > Cfg::ZERO_BYTES - 32) return kNotApplicable;
    a.lin_hpx = BM + 2 * a.W + 2;
    const long long halo_bytes = (long long)a.lin_hpx * a.lin_pitch;
    a.lin_pieces = (int)((halo_bytes + 1023) / 1024);
    long long smem = (long long)Cfg::HALO0 + (long long)a.lin_pieces * 1024;
    const long long epi = (long long)Cfg::HALO0 + (long long)BM * BN / (BN / WN) * ((BN / WN) * 2 + 16);  // the waves' staged output tiles
    if (!(MODE & 1) && epi > smem) smem = epi;
    if (smem > 160 * 1024) return kNotApplicable;
    a.div_slots = make_fastdiv((unsigned)(a.lin_pitch / 16));
    a.lin_min = a.M;  // stride 1: as many input pixels as output pixels
    const long long tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.Cout + BN - 1) / BN;
    a.nk = 9 * (a.Cin / BK);
    const long long total = tiles_m * a.tiles_n;
    if (total <= 0 || total > 0x7fffffffLL) return kNotApplicable;
    a.tiles_total = (int)total;
    static std::atomic<unsigned long long> attr_set{0};  // one bit per device: the attribute is per function and device
    const unsigned long long bit = 1ull << y5_current_device();
    if (!(attr_set.load() & bit)) {
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_lin3_kernel<BM, BN, WM, WN, NL, S, BK, MODE>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set.fetch_or(bit);
    }
    hipLaunchKernelGGL((conv_lin3_kernel<BM, BN, WM, WN, NL, S, BK, MODE>), dim3((unsigned)total), dim3(Cfg::NT), (size_t)smem, s, a);
    Y5_LAUNCH_CHECK("conv_lin3_kernel");
    return Y5_OK;
}

struct LinVariant {
    int bn;
    int (*launch)(const ConvArgs&, hipStream_t);
};
#define Y5_LIN(BM, BN, WM, WN, NL, S, BK, MODE) {BN, &launch_lin<BM, BN, WM, WN, NL, S, BK, MODE>}
const LinVariant kLin[] = {
    Y5_LIN(256, 128, 4, 2, 4, 3, 64, 0), Y5_LIN(256, 128, 4, 2, 4, 3, 64, 1), Y5_LIN(256, 128, 4, 2, 4, 3, 64, 3),
    Y5_LIN(128, 128, 2, 2, 4, 3, 64, 0), Y5_LIN(128, 128, 4, 2, 4, 3, 64, 0), Y5_LIN(128, 128, 4, 2, 4, 3, 64, 1),
    Y5_LIN(256, 64, 4, 2, 4, 3, 64, 0),  Y5_LIN(256, 64, 4, 2, 4, 3, 64, 3),  Y5_LIN(256, 64, 4, 1, 2, 3, 64, 3),
    Y5_LIN(128, 64, 2, 2, 2, 3, 64, 3),  Y5_LIN(128, 64, 4, 1, 2, 3, 64, 3),  Y5_LIN(128, 64, 4, 2, 4, 3, 64, 1),
    Y5_LIN(256, 32, 4, 1, 2, 3, 32, 3),  Y5_LIN(128, 32, 4, 1, 2, 3, 32, 3),  Y5_LIN(256, 64, 4, 2, 4, 3, 32, 3),
    Y5_LIN(256, 64, 4, 2, 4, 3, 32, 0),  Y5_LIN(256, 32, 8, 1, 4, 3, 32, 1),  Y5_LIN(256, 128, 4, 2, 4, 3, 32, 0),
};
#undef Y5_LIN

}  // namespace

#ifdef Y5_STAMPS
extern "C" int y5_debug_stamps_lin(unsigned long long* dst, int n_waves) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(y5_lin_stamp_buf), (size_t)n_waves * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif

int y5_conv_lin_count() { return (int)(sizeof(kLin) / sizeof(kLin[0])); }

// which = 1..y5_conv_lin_count().  Returns 0 when the shape is outside what this kernel family covers.
int y5_try_conv_lin(const y5_conv_desc& d, hipStream_t s, int which, int* rc) {
    if (which < 1 || which > y5_conv_lin_count()) return 0;
    if (d.ksize != 3 || d.stride != 1) return 0;
    const LinVariant& v = kLin[which - 1];
    if ((long long)((d.cout + v.bn - 1) / v.bn) * v.bn > d.w_rows) return 0;
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
    a.stride = d.stride; a.pad = d.pad; a.K = 9 * d.cin;
    a.w_ld = d.kpad;
    a.slope = d.act == Y5_ACT_LEAKY ? d.slope : 1.f;
    a.div_hw = make_fastdiv((unsigned)(d.h * d.w_in));
    a.div_wo = make_fastdiv((unsigned)d.w_in);
    const long long xb = (((long long)d.batch * d.h * d.w_in - 1) * d.x_ld + d.cin) * 2, wb = (long long)d.w_rows * d.kpad * 2;
    if (xb >= 0x7fffffffLL || wb >= 0x7fffffffLL) return 0;
    a.x_bytes = (int)xb;
    a.w_bytes = (int)wb;
#ifdef Y5_STAMPS
    if (const char* mode = getenv("Y5_LIN_MODE")) a.stats_rows = atoi(mode);
#endif
    const int r = v.launch(a, s);
    if (r == kNotApplicable) return 0;  // the tile does not fit this shape (channel count, LDS)
    *rc = r;
    return 1;
}
