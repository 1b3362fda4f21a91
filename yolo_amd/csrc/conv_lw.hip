// Implicit-GEMM convolution with dedicated loader waves ("lw"): same GEMM view, LDS ring image, swizzle, MFMA
// fragment layout and epilogue as conv_igemm.hip, but the workgroup is split by role.  WM x WN consumer waves only
// read fragments and issue MFMAs; NL loader waves only compute source offsets and issue the global->LDS DMAs
// (buffer_load ... lds).  Measured on gfx950 (tools/exp/lds_dma_rate.hip): a wave's DMA issue and its own MFMAs never
// overlap, and with global_load_lds not even those of its SIMD neighbour do, so a one-role K-step costs staging time
// plus MFMA time; buffer-descriptor DMAs issued by other waves do proceed beside the MFMAs (1550 instead of 2030-2430
// cycles for a 48 KB step next to 16 MFMAs per wave).
//
// Covers the two addressing modes that need no per-chunk coordinate decode: 3x3 (any stride; K-steps restart at every
// tap, so a step lies in one tap) and 1x1 stride 1.  Everything else stays with conv_igemm.hip.
#include "y5_common.h"

namespace {

#include "conv_common.h"
#include "decode_tile.h"

template <int BM, int BN, int WM, int WN, int NL, int S, int BK>
struct LwCfg {
    static constexpr int NT = 64 * (WM * WN + NL);
    static constexpr int STAGE_BYTES = (BM + BN) * BK * 2;
    static constexpr int EP = BN * 2 + 16;  // epilogue row pitch in bytes
    static constexpr int PRM0 = (S * STAGE_BYTES > BM * EP) ? S * STAGE_BYTES : BM * EP;
    static constexpr int SMEM = PRM0 + 3 * BN * 4;
};

// DIRECT: 1x1 stride 1 (output pixel m reads input pixel m); otherwise 3x3.
// The loaders use buffer_load ... lds: one scalar descriptor per tensor, a 32-bit byte offset per lane, zeros from the
// hardware's range check for chunks outside the image or past a tap's channels.
// DETECT: the Detect head's 1x1 convolution with the box decode as its epilogue (models/yolo.py:32-46 in one kernel):
// tiles are TP pixels of ONE image by all na*no channels, and instead of storing the NHWC tile the workgroup writes the
// permuted raw copy x[i] and the decoded rows of z from it.
// TAIL: a second, 1x1 stride-1 convolution on the finished tile (all of the first convolution's channels lie in the
// tile: tiles_n = 1): out2 = post2(W2 . out1), with out1 = the first convolution's complete fp16 result including its
// residual.  out1 goes to memory only if a.keep_first; the second GEMM reads it from LDS as its pixel operand and its
// weights straight from global memory (they are L2-resident: every workgroup reads the same rows).
template <int BM, int BN, int WM, int WN, int NL, int S, int BK, bool DIRECT, bool DETECT = false, bool TAIL = false>
__global__ __launch_bounds__(64 * (WM * WN + NL)) void conv_lw_kernel(const ConvArgs a, const DecodeArgs dec) {
    static_assert(!(DETECT && TAIL), "one special epilogue at a time");
    static_assert(!DETECT || (DIRECT && BM % TP == 0), "the decoding epilogue works on TP-pixel slices of a 1x1 convolution's tile");
    typedef LwCfg<BM, BN, WM, WN, NL, S, BK> Cfg;
    constexpr int NWC = WM * WN, NT = Cfg::NT;
    static_assert(BK == 32 || BK == 64, "K-step is 32 or 64");
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int RM = WTM / 32, RN = WTN / 32;
    static_assert(RM >= 1 && RN >= 1, "wave tile must hold at least one MFMA tile");
    constexpr int ROW_BYTES = BK * 2;
    constexpr int CPRW = BK / 8;     // 16-byte chunks per tile row
    constexpr int RPW = 64 / CPRW;   // tile rows one wave's DMA instruction covers (1 KB)
    constexpr int A_PIECES = BM / RPW, B_PIECES = BN / RPW;
    static_assert(A_PIECES % NL == 0 && B_PIECES % NL == 0, "loader waves share the stage evenly");
    constexpr int A_CH = A_PIECES / NL, B_CH = B_PIECES / NL, LPS = A_CH + B_CH;
    constexpr int STAGE_BYTES = Cfg::STAGE_BYTES, EP = Cfg::EP;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);

    const int tile = xcd_remap(blockIdx.x, a.tiles_total);
    const int mt = tile / a.tiles_n, nt = tile - mt * a.tiles_n;
    int m0 = mt * BM;
    const int n0 = nt * BN;
    int rows_valid = BM, det_b = 0, det_pix0 = 0;  // DETECT: image, first pixel and pixel count of this tile
    if constexpr (DETECT) {
        const int npix = dec.ny * dec.nx;
        det_b = mt / dec.tiles_per_image;
        det_pix0 = (mt - det_b * dec.tiles_per_image) * BM;
        rows_valid = npix - det_pix0 < BM ? npix - det_pix0 : BM;
        m0 = det_b * npix + det_pix0;
    }
    float* prm = reinterpret_cast<float*>(smem + Cfg::PRM0);
    const EpiParams epi = epi_fetch(a, n0, t, BN);
    epi_store(prm, epi, t, BN);  // published by the barriers of the K loop
    float* prm2 = prm + 3 * BN;  // TAIL: the second convolution's per-channel parameters (the launcher adds the space)
    if constexpr (TAIL) {
        if (t < BN) {
            const bool in = t < a.Cout2;
            prm2[t] = a.bias2 && in ? a.bias2[t] : 0.f;
            prm2[BN + t] = a.scale2 && in ? a.scale2[t] : 1.f;
            prm2[2 * BN + t] = a.scale2 && in ? a.shift2[t] : 0.f;
        }
    }
    const int nk = a.nk;

    if (wave >= NWC) {
        // ------------------------------------------------------------------ loader waves
        const int lw = wave - NWC;
        // Byte offsets from the tensor's first element; OOB (>= num_records under any reading of the range check,
        // and no 32-bit wrap with the offsets added to it) makes the hardware return zeros.
        constexpr unsigned OOB = 0x80000000u;
        const BufRsrc xr = make_rsrc(a.x, a.x_bytes), wr = make_rsrc(a.w, a.w_bytes);
        int abase[A_CH];       // byte offset of this lane's chunk at tap (0, 0) / K position 0 (may be negative when masked)
        unsigned vmask[A_CH];  // 3x3: validity bit per tap; 1x1: pixel in range
        int kc8[A_CH];
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const int row = (lw + NL * i) * RPW + lane / CPRW;
            const int kc = (lane % CPRW) ^ ((BK == 64) ? ((row >> 1) & 7) : ((row >> 2) & 3));
            kc8[i] = kc * 8;
            const int m = m0 + row;
            const bool mvalid = DETECT ? row < rows_valid : m < a.M;
            const unsigned mm = mvalid ? (unsigned)m : 0u;
            vmask[i] = 0;
            if (DIRECT) {
                abase[i] = (int)((mm * (unsigned)a.x_ld + kc * 8) * 2u);
                vmask[i] = mvalid ? 1u : 0u;
            } else {
                const unsigned b = fdiv(mm, a.div_hw);
                const unsigned rem = mm - b * (unsigned)(a.Ho * a.Wo);
                const unsigned oy = fdiv(rem, a.div_wo), ox = rem - oy * (unsigned)a.Wo;
                const int hi0 = (int)oy * a.stride - a.pad, wi0 = (int)ox * a.stride - a.pad;
                abase[i] = (((int)b * a.H + hi0) * a.W + wi0) * a.x_ld * 2 + kc * 16;
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    const int hi = hi0 + tp / 3, wi = wi0 + tp % 3;
                    if (mvalid && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W) vmask[i] |= 1u << tp;
                }
            }
        }
        unsigned wbase[B_CH];
#pragma unroll
        for (int i = 0; i < B_CH; ++i) {
            const int row = (lw + NL * i) * RPW + lane / CPRW;
            const int kc = (lane % CPRW) ^ ((BK == 64) ? ((row >> 1) & 7) : ((row >> 2) & 3));
            wbase[i] = ((unsigned)(n0 + row) * (unsigned)a.w_ld + kc * 8) * 2u;
        }
        int tap = 0, c0 = 0, kbase = 0;
        auto issue = [&](int stage) {
            int off = kbase * 2;
            unsigned bit = 1u;
            if (!DIRECT) {
                const int r = (tap * 11) >> 5, q = tap - 3 * r;
                off = ((r * a.W + q) * a.x_ld + c0) * 2;
                bit = 1u << tap;
            }
            const int koff = (DIRECT ? kbase : tap * a.Cin + c0) * 2;  // this K-step's byte offset in a weight row
            unsigned char* base = smem + stage * STAGE_BYTES + lw * 1024;
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                const bool ok = DIRECT ? (vmask[i] && kbase + kc8[i] < a.K) : ((vmask[i] & bit) != 0 && c0 + kc8[i] < a.Cin);
                const unsigned vo = ok ? (unsigned)(abase[i] + off) : OOB;
                buf_dma16(xr, (lds_ptr)(base + i * (NL * 1024)), vo, 0);
            }
#pragma unroll
            for (int i = 0; i < B_CH; ++i)
                buf_dma16(wr, (lds_ptr)(base + BM * ROW_BYTES + i * (NL * 1024)), wbase[i] + (unsigned)koff, 0);
            kbase += BK;
            if (!DIRECT) {
                c0 += BK;
                if (c0 >= a.Cin) {
                    c0 = 0;
                    ++tap;
                }
            }
        };
#pragma unroll
        for (int s = 0; s < S - 1; ++s)
            if (s < nk) issue(s);
        for (int kt = 0; kt < nk; ++kt) {
            int ahead = nk - 1 - kt;
            if (ahead > S - 2) ahead = S - 2;
            wait_vm_and_barrier(ahead * LPS);
            if (kt + S - 1 < nk) issue((kt + S - 1) % S);
        }
        wait_vm_and_barrier(0);
    } else {
        // ------------------------------------------------------------------ consumer waves
        const int wm = wave % WM, wn = wave / WM;
        const int lr = lane & 31, lh = lane >> 5;
        const int sw = (BK == 64) ? ((lr >> 1) & 7) : ((lr >> 2) & 3);  // wave tile rows start at multiples of 32
        f32x16 acc[RN][RM];
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;
        for (int kt = 0; kt < nk; ++kt) {
            wait_vm_and_barrier(0);  // own fragment reads of step kt-1 are done (lgkmcnt 0); step kt has landed
            const unsigned char* A = smem + (kt % S) * STAGE_BYTES;
            const unsigned char* B = A + BM * ROW_BYTES;
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                const int coff = ((2 * ks + lh) ^ sw) << 4;
                f16x8 xf[RM], wf[RN];
#pragma unroll
                for (int i = 0; i < RM; ++i)
                    xf[i] = *reinterpret_cast<const f16x8*>(A + (wm * WTM + i * 32 + lr) * ROW_BYTES + coff);
#pragma unroll
                for (int j = 0; j < RN; ++j)
                    wf[j] = *reinterpret_cast<const f16x8*>(B + (wn * WTN + j * 32 + lr) * ROW_BYTES + coff);
#pragma unroll
                for (int j = 0; j < RN; ++j)
#pragma unroll
                    for (int i = 0; i < RM; ++i)
                        acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j], xf[i], acc[j][i], 0, 0, 0);
            }
        }
        wait_vm_and_barrier(0);  // every consumer is done reading the ring before the epilogue reuses it
        // post-ops in registers -> fp16 tile in LDS.  accumulator element e of MFMA tile (j, i): channel =
        // 8*(e>>2) + 4*lh + (e&3) within the 32-channel tile, pixel = lr.
        const bool has_bias = a.bias != nullptr, has_scale = a.scale != nullptr;
#pragma unroll
        for (int j = 0; j < RN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = wn * WTN + j * 32 + 8 * g + 4 * lh;
                if (TAIL || n0 + nl < a.Cout) {  // TAIL: columns past Cout are written too (zeros: zero weight rows)
#pragma unroll
                    for (int i = 0; i < RM; ++i) {
                        const f32x4 v = {acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]};
                        const int ml = wm * WTM + i * 32 + lr;
                        *reinterpret_cast<f16x4*>(smem + ml * EP + nl * 2) = post_op4_lds(v, nl, prm, BN, has_bias, has_scale, a.slope);
                    }
                }
            }
        }
    }
    __syncthreads();
    if constexpr (DETECT) {
        f16* tile = reinterpret_cast<f16*>(smem);
        f16* boxdec = reinterpret_cast<f16*>(smem + Cfg::SMEM);  // the launcher adds DETECT_EXTRA bytes behind the parameters
#pragma unroll
        for (int sub = 0; sub < BM / TP; ++sub) {
            const int np = rows_valid - sub * TP < TP ? rows_valid - sub * TP : TP;  // workgroup-uniform
            if (np <= 0) break;
            if (sub) __syncthreads();  // the box slots of the previous slice are consumed
            decode_from_lds<true, NT>(dec, det_b, det_pix0 + sub * TP, np, tile + sub * TP * (EP / 2), boxdec);
        }
        return;
    }
    constexpr int CPR = BN / 8;    // 16-byte chunks per tile row
    constexpr int RPP = NT / CPR;  // rows per pass
    static_assert(NT % CPR == 0, "whole rows per pass");
    const int cc = t % CPR, rr = t / CPR;
    const int n = n0 + cc * 8;
    if constexpr (TAIL) {
        // ---- A: the residual goes into the tile (and the first output to memory when it is kept)
        if (a.res || a.keep_first) {
            if (n < a.Cout) {
#pragma unroll
                for (int p = 0; p < (BM + RPP - 1) / RPP; ++p) {
                    const int row = rr + p * RPP;
                    const int m = m0 + row;
                    if (row < BM && m < a.M) {
                        f16x8 v = *reinterpret_cast<const f16x8*>(smem + row * EP + cc * 16);
                        if (a.res) {
                            v += *reinterpret_cast<const f16x8*>(a.res + (size_t)m * a.res_ld + n);
                            *reinterpret_cast<f16x8*>(smem + row * EP + cc * 16) = v;
                        }
                        if (a.keep_first) *reinterpret_cast<f16x8*>(a.y + (size_t)m * a.y_ld + n) = v;
                    }
                }
            }
            __syncthreads();
        }
        // ---- B: second GEMM, pixels x first-output channels from the tile, weights from global memory
        const int wm = wave % WM, wn = wave / WM;
        const int lr = lane & 31, lh = lane >> 5;
        f32x16 acc2[RN][RM];
        const bool mine = wave < NWC && wn * WTN < a.Cout2;  // wave-uniform: this wave owns second-output channels
        if (mine) {
#pragma unroll
            for (int j = 0; j < RN; ++j)
#pragma unroll
                for (int i = 0; i < RM; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc2[j][i][e] = 0.f;
            const int k2steps = (a.Cout + 15) / 16;  // the tile holds zeros from Cout up to BN
#pragma unroll 2
            for (int ks = 0; ks < k2steps; ++ks) {
                const int k = ks * 16 + lh * 8;
                f16x8 xf[RM], wf[RN];
#pragma unroll
                for (int j = 0; j < RN; ++j)
                    wf[j] = *reinterpret_cast<const f16x8*>(a.w2 + (size_t)(wn * WTN + j * 32 + lr) * a.w2_ld + k);
#pragma unroll
                for (int i = 0; i < RM; ++i)
                    xf[i] = *reinterpret_cast<const f16x8*>(smem + (wm * WTM + i * 32 + lr) * EP + k * 2);
#pragma unroll
                for (int j = 0; j < RN; ++j)
#pragma unroll
                    for (int i = 0; i < RM; ++i)
                        acc2[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j], xf[i], acc2[j][i], 0, 0, 0);
            }
        }
        __syncthreads();  // every wave has read its rows of the first output
        if (mine) {
            const bool has_bias2 = a.bias2 != nullptr, has_scale2 = a.scale2 != nullptr;
#pragma unroll
            for (int j = 0; j < RN; ++j) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = wn * WTN + j * 32 + 8 * g + 4 * lh;
                    if (nl < a.Cout2) {
#pragma unroll
                        for (int i = 0; i < RM; ++i) {
                            const f32x4 v = {acc2[j][i][4 * g], acc2[j][i][4 * g + 1], acc2[j][i][4 * g + 2], acc2[j][i][4 * g + 3]};
                            const int ml = wm * WTM + i * 32 + lr;
                            *reinterpret_cast<f16x4*>(smem + ml * EP + nl * 2) = post_op4_lds(v, nl, prm2, BN, has_bias2, has_scale2, a.slope2);
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (n < a.Cout2) {
#pragma unroll
            for (int p = 0; p < (BM + RPP - 1) / RPP; ++p) {
                const int row = rr + p * RPP;
                const int m = m0 + row;
                if (row < BM && m < a.M)
                    *reinterpret_cast<f16x8*>(a.y2 + (size_t)m * a.y2_ld + n) = *reinterpret_cast<const f16x8*>(smem + row * EP + cc * 16);
            }
        }
        return;
    }
    // ---- all waves: 16-byte stores of the finished tile (+ residual)
    if (n < a.Cout) {
#pragma unroll
        for (int p = 0; p < (BM + RPP - 1) / RPP; ++p) {
            const int row = rr + p * RPP;
            const int m = m0 + row;
            if (row < BM && m < a.M) {
                f16x8 v = *reinterpret_cast<const f16x8*>(smem + row * EP + cc * 16);
                if (a.res) v += *reinterpret_cast<const f16x8*>(a.res + (size_t)m * a.res_ld + n);
                *reinterpret_cast<f16x8*>(a.y + (size_t)m * a.y_ld + n) = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Persistent form of the plain convolution above: a workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...  Its
// loader waves run one tile ahead of the epilogue: as soon as the consumers have read the last K-step of tile i
// (barrier F) the loaders fill the first ring stages of tile i+1 while the consumers apply the post-ops, stage the
// tile in its own LDS region (not the ring) and store it.  Only consumer waves store.  Barrier sequence per tile, the
// same for both roles: nk K-step barriers, F, E (tile staged); a loader passes E of tile i after it has issued the
// prologue of tile i+1, without waiting for those DMAs.
template <int BM, int BN, int WM, int WN, int NL, int S, int BK>
struct LwpCfg {
    typedef LwCfg<BM, BN, WM, WN, NL, S, BK> Base;
    static constexpr int EPI0 = S * Base::STAGE_BYTES;          // staged output tile, behind the ring
    static constexpr int PRM0 = EPI0 + BM * Base::EP;           // per-channel parameters
    static constexpr int SMEM = PRM0 + 3 * BN * 4;
};

__device__ __forceinline__ void wait_lgkm_and_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int BM, int BN, int WM, int WN, int NL, int S, int BK, bool DIRECT>
__global__ __launch_bounds__(64 * (WM * WN + NL)) void conv_lwp_kernel(const ConvArgs a) {
    typedef LwCfg<BM, BN, WM, WN, NL, S, BK> Cfg;
    typedef LwpCfg<BM, BN, WM, WN, NL, S, BK> PCfg;
    constexpr int NWC = WM * WN;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int RM = WTM / 32, RN = WTN / 32;
    constexpr int ROW_BYTES = BK * 2;
    constexpr int CPRW = BK / 8;
    constexpr int RPW = 64 / CPRW;
    constexpr int A_PIECES = BM / RPW, B_PIECES = BN / RPW;
    static_assert(A_PIECES % NL == 0 && B_PIECES % NL == 0, "loader waves share the stage evenly");
    constexpr int A_CH = A_PIECES / NL, B_CH = B_PIECES / NL, LPS = A_CH + B_CH;
    constexpr int STAGE_BYTES = Cfg::STAGE_BYTES, EP = Cfg::EP;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* etile = smem + PCfg::EPI0;
    float* prm = reinterpret_cast<float*>(smem + PCfg::PRM0);

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nk = a.nk;
    const int ntiles = a.tiles_total;

    if (wave >= NWC) {
        // ------------------------------------------------------------------ loader waves
        const int lw = wave - NWC;
        constexpr unsigned OOB = 0x80000000u;
        const BufRsrc xr = make_rsrc(a.x, a.x_bytes), wr = make_rsrc(a.w, a.w_bytes);
        bool first = true;
        for (int it = blockIdx.x; it < ntiles; it += gridDim.x) {
            const int tile = xcd_remap(it, ntiles);
            const int mt = tile / a.tiles_n, nt = tile - mt * a.tiles_n;
            const int m0 = mt * BM, n0 = nt * BN;
            int abase[A_CH];
            unsigned vmask[A_CH];
            int kc8[A_CH];
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                const int row = (lw + NL * i) * RPW + lane / CPRW;
                const int kc = (lane % CPRW) ^ ((BK == 64) ? ((row >> 1) & 7) : ((row >> 2) & 3));
                kc8[i] = kc * 8;
                const int m = m0 + row;
                const bool mvalid = m < a.M;
                const unsigned mm = mvalid ? (unsigned)m : 0u;
                vmask[i] = 0;
                if (DIRECT) {
                    abase[i] = (int)((mm * (unsigned)a.x_ld + kc * 8) * 2u);
                    vmask[i] = mvalid ? 1u : 0u;
                } else {
                    const unsigned b = fdiv(mm, a.div_hw);
                    const unsigned rem = mm - b * (unsigned)(a.Ho * a.Wo);
                    const unsigned oy = fdiv(rem, a.div_wo), ox = rem - oy * (unsigned)a.Wo;
                    const int hi0 = (int)oy * a.stride - a.pad, wi0 = (int)ox * a.stride - a.pad;
                    abase[i] = (((int)b * a.H + hi0) * a.W + wi0) * a.x_ld * 2 + kc * 16;
#pragma unroll
                    for (int tp = 0; tp < 9; ++tp) {
                        const int hi = hi0 + tp / 3, wi = wi0 + tp % 3;
                        if (mvalid && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W) vmask[i] |= 1u << tp;
                    }
                }
            }
            unsigned wbase[B_CH];
#pragma unroll
            for (int i = 0; i < B_CH; ++i) {
                const int row = (lw + NL * i) * RPW + lane / CPRW;
                const int kc = (lane % CPRW) ^ ((BK == 64) ? ((row >> 1) & 7) : ((row >> 2) & 3));
                wbase[i] = ((unsigned)(n0 + row) * (unsigned)a.w_ld + kc * 8) * 2u;
            }
            int tap = 0, c0 = 0, kbase = 0;
            auto issue = [&](int stage) {
                int off = kbase * 2;
                unsigned bit = 1u;
                if (!DIRECT) {
                    const int r = (tap * 11) >> 5, q = tap - 3 * r;
                    off = ((r * a.W + q) * a.x_ld + c0) * 2;
                    bit = 1u << tap;
                }
                const int koff = (DIRECT ? kbase : tap * a.Cin + c0) * 2;
                unsigned char* base = smem + stage * STAGE_BYTES + lw * 1024;
#pragma unroll
                for (int i = 0; i < A_CH; ++i) {
                    const bool ok = DIRECT ? (vmask[i] && kbase + kc8[i] < a.K) : ((vmask[i] & bit) != 0 && c0 + kc8[i] < a.Cin);
                    const unsigned vo = ok ? (unsigned)(abase[i] + off) : OOB;
                    buf_dma16(xr, (lds_ptr)(base + i * (NL * 1024)), vo, 0);
                }
#pragma unroll
                for (int i = 0; i < B_CH; ++i)
                    buf_dma16(wr, (lds_ptr)(base + BM * ROW_BYTES + i * (NL * 1024)), wbase[i] + (unsigned)koff, 0);
                kbase += BK;
                if (!DIRECT) {
                    c0 += BK;
                    if (c0 >= a.Cin) {
                        c0 = 0;
                        ++tap;
                    }
                }
            };
            // the ring is free: every consumer passed barrier F of the previous tile
#pragma unroll
            for (int s = 0; s < S - 1; ++s)
                if (s < nk) issue(s);
            if (!first) wait_vm_and_barrier((S - 1) * LPS);  // barrier E of the previous tile; the prologue stays in flight
            first = false;
            for (int kt = 0; kt < nk; ++kt) {
                int ahead = nk - 1 - kt;
                if (ahead > S - 2) ahead = S - 2;
                wait_vm_and_barrier(ahead * LPS);
                if (kt + S - 1 < nk) issue((kt + S - 1) % S);
            }
            wait_vm_and_barrier(0);  // F
        }
        if (!first) wait_vm_and_barrier(0);  // barrier E of the last tile
        return;
    }

    // ---------------------------------------------------------------------- consumer waves
    const int wm = wave % WM, wn = wave / WM;
    const int lr = lane & 31, lh = lane >> 5;
    const int sw = (BK == 64) ? ((lr >> 1) & 7) : ((lr >> 2) & 3);
    const bool has_bias = a.bias != nullptr, has_scale = a.scale != nullptr;
    constexpr int NTC = 64 * NWC;  // consumer threads: they alone stage and store the tile
    constexpr int CPR = BN / 8;
    constexpr int RPP = NTC / CPR;
    static_assert(NTC % CPR == 0 && NTC >= BN, "consumer threads cover whole tile rows and the tile's channels");
    const int cc = t % CPR, rr = t / CPR;
    for (int it = blockIdx.x; it < ntiles; it += gridDim.x) {
        const int tile = xcd_remap(it, ntiles);
        const int mt = tile / a.tiles_n, nt = tile - mt * a.tiles_n;
        const int m0 = mt * BM, n0 = nt * BN;
        // this tile's per-channel parameters: the previous tile's were last read before its barrier E; published by the
        // K-step barriers below
        epi_store(prm, epi_fetch(a, n0, t, BN), t, BN);
        f32x16 acc[RN][RM];
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;
        for (int kt = 0; kt < nk; ++kt) {
            wait_lgkm_and_barrier();  // own fragment reads of step kt-1 are done; the loaders saw step kt land
            const unsigned char* A = smem + (kt % S) * STAGE_BYTES;
            const unsigned char* B = A + BM * ROW_BYTES;
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                const int coff = ((2 * ks + lh) ^ sw) << 4;
                f16x8 xf[RM], wf[RN];
#pragma unroll
                for (int i = 0; i < RM; ++i)
                    xf[i] = *reinterpret_cast<const f16x8*>(A + (wm * WTM + i * 32 + lr) * ROW_BYTES + coff);
#pragma unroll
                for (int j = 0; j < RN; ++j)
                    wf[j] = *reinterpret_cast<const f16x8*>(B + (wn * WTN + j * 32 + lr) * ROW_BYTES + coff);
#pragma unroll
                for (int j = 0; j < RN; ++j)
#pragma unroll
                    for (int i = 0; i < RM; ++i)
                        acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j], xf[i], acc[j][i], 0, 0, 0);
            }
        }
        wait_lgkm_and_barrier();  // F: every consumer is done reading the ring
#pragma unroll
        for (int j = 0; j < RN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = wn * WTN + j * 32 + 8 * g + 4 * lh;
                if (n0 + nl < a.Cout) {
#pragma unroll
                    for (int i = 0; i < RM; ++i) {
                        const f32x4 v = {acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]};
                        const int ml = wm * WTM + i * 32 + lr;
                        *reinterpret_cast<f16x4*>(etile + ml * EP + nl * 2) = post_op4_lds(v, nl, prm, BN, has_bias, has_scale, a.slope);
                    }
                }
            }
        }
        wait_lgkm_and_barrier();  // E: the tile is staged
        const int n = n0 + cc * 8;
        if (n < a.Cout) {
#pragma unroll
            for (int p = 0; p < (BM + RPP - 1) / RPP; ++p) {
                const int row = rr + p * RPP;
                const int m = m0 + row;
                if (row < BM && m < a.M) {
                    f16x8 v = *reinterpret_cast<const f16x8*>(etile + row * EP + cc * 16);
                    if (a.res) v += *reinterpret_cast<const f16x8*>(a.res + (size_t)m * a.res_ld + n);
                    *reinterpret_cast<f16x8*>(a.y + (size_t)m * a.y_ld + n) = v;
                }
            }
        }
    }
}

int y5_cu_count() {
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

template <int BM, int BN, int WM, int WN, int NL, int S, int BK>
int launch_lwp(const ConvArgs& a0, bool direct, bool tail, hipStream_t s) {
    typedef LwpCfg<BM, BN, WM, WN, NL, S, BK> PCfg;
    static_assert(PCfg::SMEM <= 160 * 1024, "ring + staged tile must fit the CU's LDS");
    if (tail) return y5_fail(Y5_EINVAL, "conv chain: the persistent variants run single convolutions only");
    ConvArgs a = a0;
    const long long tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.Cout + BN - 1) / BN;
    a.nk = direct ? (a.K + BK - 1) / BK : 9 * ((a.Cin + BK - 1) / BK);
    const long long total = tiles_m * a.tiles_n;
    if (total <= 0 || total > 0x7fffffffLL) return y5_fail(Y5_EINVAL, "conv: tile count %lld out of range", total);
    a.tiles_total = (int)total;
    static std::atomic<unsigned long long> attr_set{0};
    const unsigned long long bit = 1ull << y5_current_device();
    if (!(attr_set.load() & bit)) {
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_lwp_kernel<BM, BN, WM, WN, NL, S, BK, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, PCfg::SMEM));
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_lwp_kernel<BM, BN, WM, WN, NL, S, BK, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, PCfg::SMEM));
        attr_set.fetch_or(bit);
    }
    // as many workgroups as stay resident at once (LDS and the 2048 threads of a CU), each walking its share of tiles
    constexpr int NT = 64 * (WM * WN + NL);
    int per_cu = (160 * 1024) / PCfg::SMEM;
    if (per_cu > 2048 / NT) per_cu = 2048 / NT;
    if (per_cu < 1) per_cu = 1;
    long long grid = (long long)y5_cu_count() * per_cu;
    if (grid > total) grid = total;
    if (direct)
        hipLaunchKernelGGL((conv_lwp_kernel<BM, BN, WM, WN, NL, S, BK, true>), dim3((unsigned)grid), dim3(NT), PCfg::SMEM, s, a);
    else
        hipLaunchKernelGGL((conv_lwp_kernel<BM, BN, WM, WN, NL, S, BK, false>), dim3((unsigned)grid), dim3(NT), PCfg::SMEM, s, a);
    Y5_LAUNCH_CHECK("conv_lwp_kernel");
    return Y5_OK;
}

template <int BM, int BN, int WM, int WN, int NL, int S, int BK>
int launch_lw(const ConvArgs& a0, bool direct, bool tail, hipStream_t s) {
    typedef LwCfg<BM, BN, WM, WN, NL, S, BK> Cfg;
    ConvArgs a = a0;
    const long long tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.Cout + BN - 1) / BN;
    // 1x1: K-steps over the weight row, zero-padded to a multiple of 64 >= K.  3x3: every tap starts a fresh K-step, so a
    // step never mixes taps; where Cin is not a multiple of BK the tap's last step stages zeros for the missing channels
    // (and whatever weights follow: their products vanish).
    a.nk = direct ? (a.K + BK - 1) / BK : 9 * ((a.Cin + BK - 1) / BK);
    const long long total = tiles_m * a.tiles_n;
    if (total <= 0 || total > 0x7fffffffLL) return y5_fail(Y5_EINVAL, "conv: tile count %lld out of range", total);
    a.tiles_total = (int)total;
    constexpr int SMEM_TAIL = Cfg::SMEM + 3 * BN * 4;  // + the second convolution's per-channel parameters
    static std::atomic<unsigned long long> attr_set{0};  // one bit per device: the attribute is per function and device
    const unsigned long long bit = 1ull << y5_current_device();
    if (!(attr_set.load() & bit)) {
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_lw_kernel<BM, BN, WM, WN, NL, S, BK, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_lw_kernel<BM, BN, WM, WN, NL, S, BK, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_lw_kernel<BM, BN, WM, WN, NL, S, BK, false, false, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_TAIL));
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_lw_kernel<BM, BN, WM, WN, NL, S, BK, true, false, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_TAIL));
        attr_set.fetch_or(bit);
    }
    const dim3 grid((unsigned)total), block(Cfg::NT);
    if (tail && a.tiles_n != 1) return y5_fail(Y5_EINVAL, "conv chain: the first convolution's %d channels exceed the %d-channel tile", a.Cout, BN);
    if (tail && direct)
        hipLaunchKernelGGL((conv_lw_kernel<BM, BN, WM, WN, NL, S, BK, true, false, true>), grid, block, SMEM_TAIL, s, a, DecodeArgs{});
    else if (tail)
        hipLaunchKernelGGL((conv_lw_kernel<BM, BN, WM, WN, NL, S, BK, false, false, true>), grid, block, SMEM_TAIL, s, a, DecodeArgs{});
    else if (direct)
        hipLaunchKernelGGL((conv_lw_kernel<BM, BN, WM, WN, NL, S, BK, true>), grid, block, Cfg::SMEM, s, a, DecodeArgs{});
    else
        hipLaunchKernelGGL((conv_lw_kernel<BM, BN, WM, WN, NL, S, BK, false>), grid, block, Cfg::SMEM, s, a, DecodeArgs{});
    Y5_LAUNCH_CHECK("conv_lw_kernel");
    return Y5_OK;
}

struct LwVariant {
    int bn, bk;
    int (*launch)(const ConvArgs&, bool, bool, hipStream_t);
};
#define Y5_LW(BM, BN, WM, WN, NL, S, BK) {BN, BK, &launch_lw<BM, BN, WM, WN, NL, S, BK>}
#define Y5_LWP(BM, BN, WM, WN, NL, S, BK) {BN, BK, &launch_lwp<BM, BN, WM, WN, NL, S, BK>}
const LwVariant kLw[] = {
    Y5_LW(256, 128, 4, 2, 8, 2, 64), Y5_LW(256, 128, 4, 2, 8, 3, 64), Y5_LW(256, 128, 4, 2, 8, 3, 32),
    Y5_LW(128, 128, 2, 2, 4, 2, 64), Y5_LW(128, 128, 2, 2, 4, 3, 32), Y5_LW(128, 128, 2, 4, 8, 3, 64),
    Y5_LW(256, 64, 4, 2, 8, 3, 64),  Y5_LW(128, 64, 2, 2, 4, 3, 64),  Y5_LW(64, 64, 2, 2, 4, 3, 64),
    Y5_LW(64, 128, 2, 2, 4, 3, 64),  Y5_LW(256, 256, 2, 4, 4, 2, 64), Y5_LW(256, 256, 2, 4, 4, 3, 32),
    // persistent workgroups (conv_lwp_kernel): single convolutions only
    Y5_LWP(256, 128, 4, 2, 8, 3, 32), Y5_LWP(128, 128, 2, 2, 4, 2, 64), Y5_LWP(128, 128, 2, 2, 4, 3, 32),
    Y5_LWP(128, 128, 2, 4, 8, 3, 64), Y5_LWP(128, 64, 2, 2, 4, 3, 64),  Y5_LWP(64, 64, 2, 2, 4, 3, 64),
    Y5_LWP(256, 64, 4, 2, 8, 3, 64),  Y5_LWP(64, 128, 2, 2, 4, 3, 64),
};
#undef Y5_LW
#undef Y5_LWP

constexpr int DETECT_EXTRA = (TP + 1) * Y5_MAX_ANCHORS * 4 * (int)sizeof(f16);  // decoded box slots behind the parameters

template <int BM, int WM, int WN, int NL, int S, int BK>
int launch_detect(const ConvArgs& a0, const DecodeArgs& dec0, int batch, hipStream_t s) {
    constexpr int BN = 256;
    typedef LwCfg<BM, BN, WM, WN, NL, S, BK> Cfg;
    ConvArgs a = a0;
    DecodeArgs dec = dec0;
    dec.lds_ld = Cfg::EP / 2;  // the conv epilogue's tile pitch
    dec.tiles_per_image = (dec.ny * dec.nx + BM - 1) / BM;  // tiles of BM pixels of one image here
    a.tiles_n = 1;
    a.nk = (a.K + BK - 1) / BK;
    const long long total = (long long)dec.tiles_per_image * batch;
    if (total <= 0 || total > 0x7fffffffLL) return y5_fail(Y5_EINVAL, "detect: tile count %lld out of range", total);
    a.tiles_total = (int)total;
    static std::atomic<unsigned long long> attr_set{0};
    const unsigned long long bit = 1ull << y5_current_device();
    if (!(attr_set.load() & bit)) {
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_lw_kernel<BM, BN, WM, WN, NL, S, BK, true, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM + DETECT_EXTRA));
        attr_set.fetch_or(bit);
    }
    hipLaunchKernelGGL((conv_lw_kernel<BM, BN, WM, WN, NL, S, BK, true, true>), dim3((unsigned)total), dim3(Cfg::NT), Cfg::SMEM + DETECT_EXTRA, s, a, dec);
    Y5_LAUNCH_CHECK("conv_lw_kernel (detect)");
    return Y5_OK;
}

typedef int (*DetectLaunch)(const ConvArgs&, const DecodeArgs&, int, hipStream_t);
const DetectLaunch kDetect[] = {&launch_detect<64, 2, 4, 4, 2, 32>, &launch_detect<64, 2, 4, 4, 3, 32>, &launch_detect<64, 2, 2, 4, 3, 32>,
                                &launch_detect<64, 2, 4, 8, 3, 64>, &launch_detect<64, 2, 2, 2, 2, 32>, &launch_detect<64, 2, 2, 4, 2, 32>,
                                &launch_detect<64, 1, 4, 4, 2, 32>};

}  // namespace

// the tile configurations of this file + the persistent kernel of conv_detect.hip as the last variant
int y5_detect_variant_count() { return (int)(sizeof(kDetect) / sizeof(kDetect[0])) + 1; }

// Detect level = 1x1 convolution (bias, no activation) + decode, one kernel.  conv.y is not written.
int y5_launch_detect(const y5_conv_desc& c, const y5_decode_desc& d, int variant, hipStream_t s) {
    Y5_REQUIRE(c.dtype == Y5_F16 && d.dtype == Y5_F16, "detect: fp16 only");
    Y5_REQUIRE(variant >= 0 && variant <= y5_detect_variant_count(), "detect: variant %d out of range 0..%d", variant, y5_detect_variant_count());
    Y5_REQUIRE(c.x && c.w && d.z_out, "detect: null tensor pointer");
    Y5_REQUIRE(c.ksize == 1 && c.stride == 1 && c.pad == 0, "detect: the head convolution is 1x1 stride 1");
    Y5_REQUIRE(c.batch == d.batch && c.h == d.ny && c.w_in == d.nx && c.ho == d.ny && c.wo == d.nx, "detect: convolution and decode shapes differ");
    Y5_REQUIRE(c.cin > 0 && c.cin % 8 == 0 && c.x_ld % 8 == 0 && c.x_ld >= c.cin, "detect: cin %d / x_ld %d must be multiples of 8", c.cin, c.x_ld);
    Y5_REQUIRE(c.cout % 8 == 0 && c.cout >= d.na * d.no && c.cout <= 256, "detect: %d output channels for na*no = %d (at most 256)", c.cout, d.na * d.no);
    Y5_REQUIRE(c.kpad % 64 == 0 && c.kpad >= c.cin && c.w_rows >= 256, "detect: weights must be packed to 256 rows of kpad >= cin");
    Y5_REQUIRE(!c.res && !c.scale && c.act == Y5_ACT_NONE, "detect: the head convolution has a bias only");
    Y5_REQUIRE(y5_aligned16(c.x) && y5_aligned16(c.w) && y5_aligned16(c.bias), "detect: pointers must be 16-byte aligned");
    DecodeArgs dec;
    size_t lds_bytes = 0;
    bool vec = false;
    if (int rc = prepare_level(d, dec, &lds_bytes, &vec)) return rc;
    Y5_REQUIRE(vec, "detect: the fused kernel needs 16-byte aligned output runs (ny*nx*no, rows*no, row0*no multiples of 8, no >= 6)");
    const long long xb = (((long long)c.batch * c.h * c.w_in - 1) * c.x_ld + c.cin) * 2, wb = (long long)c.w_rows * c.kpad * 2;
    Y5_REQUIRE(xb < 0x7fffffffLL && wb < 0x7fffffffLL, "detect: tensors of 2 GiB or more");
    ConvArgs a{};
    a.x = static_cast<const f16*>(c.x);
    a.w = static_cast<const f16*>(c.w);
    a.bias = c.bias;
    a.M = (long long)c.batch * c.h * c.w_in;
    a.H = c.h; a.W = c.w_in; a.Cin = c.cin; a.x_ld = c.x_ld;
    a.Ho = c.h; a.Wo = c.w_in; a.Cout = c.cout;
    a.stride = 1; a.pad = 0; a.K = c.cin;
    a.w_ld = c.kpad;
    a.slope = 1.f;
    a.x_bytes = (int)xb;
    a.w_bytes = (int)wb;
    if (variant == y5_detect_variant_count()) {
        int rc = Y5_OK;
        if (y5_try_detect_persistent(c, d, s, &rc)) return rc;
        return y5_fail(Y5_EINVAL, "detect: the persistent kernel takes heads of 128 input channels");
    }
    return kDetect[variant > 0 ? variant - 1 : 0](a, dec, c.batch, s);
}

int y5_conv_lw_count() { return (int)(sizeof(kLw) / sizeof(kLw[0])); }

namespace {
// Kernel arguments of a convolution this family covers; false when it does not (addressing mode, tensor spans).
bool fill_args(const y5_conv_desc& d, ConvArgs& a, bool* direct) {
    *direct = d.ksize == 1 && d.stride == 1;
    if (!*direct && d.ksize != 3) return false;
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
    a.w_ld = d.kpad;
    a.slope = d.act == Y5_ACT_LEAKY ? d.slope : 1.f;
    a.div_hw = make_fastdiv((unsigned)(d.ho * d.wo));
    a.div_wo = make_fastdiv((unsigned)d.wo);
    a.div_cin = make_fastdiv((unsigned)d.cin);
    // spans the loaders may touch, for the buffer descriptors (32-bit offsets, sentinel 2^31 for zeros)
    const long long xb = (((long long)d.batch * d.h * d.w_in - 1) * d.x_ld + d.cin) * 2, wb = (long long)d.w_rows * d.kpad * 2;
    if (xb >= 0x7fffffffLL || wb >= 0x7fffffffLL) return false;
    a.x_bytes = (int)xb;
    a.w_bytes = (int)wb;
    return true;
}
}  // namespace

// which = 1..y5_conv_lw_count().  Returns 0 when the shape is outside what this kernel family covers.
int y5_try_conv_lw(const y5_conv_desc& d, hipStream_t s, int which, int* rc) {
    if (which < 1 || which > y5_conv_lw_count()) return 0;
    const LwVariant& v = kLw[which - 1];
    if ((long long)((d.cout + v.bn - 1) / v.bn) * v.bn > d.w_rows) return 0;
    ConvArgs a{};
    bool direct = false;
    if (!fill_args(d, a, &direct)) return 0;
    *rc = v.launch(a, direct, false, s);
    return 1;
}

// A convolution followed by a 1x1 stride-1 convolution of its output, in one kernel (conv_lw_kernel, TAIL).
// variant: 0 = the first tile shape that holds both channel counts, else 1..y5_conv_lw_count().
int y5_launch_conv_chain(const y5_conv_desc& f, const y5_conv_desc& g, int keep_first, int variant, hipStream_t s) {
    Y5_REQUIRE(f.dtype == Y5_F16 && g.dtype == Y5_F16, "conv chain: fp16 only");
    Y5_REQUIRE(variant >= 0 && variant <= y5_conv_lw_count() + 3, "conv chain: variant %d out of range 0..%d", variant, y5_conv_lw_count() + 3);
    Y5_REQUIRE(f.x && f.w && g.w && g.y && (f.y || !keep_first), "conv chain: null tensor pointer");
    Y5_REQUIRE(f.ksize == 3 || (f.ksize == 1 && f.stride == 1), "conv chain: the first convolution is 3x3 or 1x1 stride 1");
    Y5_REQUIRE(f.stride == 1 || f.stride == 2, "conv chain: stride %d not in {1,2}", f.stride);
    Y5_REQUIRE(f.pad == f.ksize / 2 && f.batch > 0 && f.ho == (f.h + 2 * f.pad - f.ksize) / f.stride + 1 && f.wo == (f.w_in + 2 * f.pad - f.ksize) / f.stride + 1,
               "conv chain: inconsistent geometry of the first convolution");
    Y5_REQUIRE(f.cin > 0 && f.cin % 8 == 0 && f.x_ld % 8 == 0 && f.x_ld >= f.cin && f.cout > 0 && f.cout % 8 == 0, "conv chain: channel counts must be multiples of 8");
    Y5_REQUIRE(!keep_first || (f.y_ld % 8 == 0 && f.y_ld >= f.cout), "conv chain: bad pitch of the first output");
    Y5_REQUIRE(f.kpad % 64 == 0 && f.kpad >= f.ksize * f.ksize * f.cin && f.w_rows >= f.cout, "conv chain: first weights not packed");
    Y5_REQUIRE(!f.scale || f.shift, "conv chain: scale without shift");
    Y5_REQUIRE(!f.res || (f.res_ld % 8 == 0 && f.res_ld >= f.cout), "conv chain: bad residual pitch %d", f.res_ld);
    Y5_REQUIRE(g.ksize == 1 && g.stride == 1 && g.pad == 0 && !g.res, "conv chain: the second convolution is 1x1 stride 1 without residual");
    // the second convolution reads the first one's output - all of its channels, or (persistent 1x1 kernel only) the leading g.cin
    const bool part = g.cin < f.cout;
    Y5_REQUIRE(g.batch == f.batch && g.h == f.ho && g.w_in == f.wo && g.ho == f.ho && g.wo == f.wo && g.cin > 0 && g.cin <= f.cout,
               "conv chain: the second convolution must take the first one's output");
    Y5_REQUIRE(!part || variant == 0 || variant == y5_conv_lw_count() + 2, "conv chain: variant %d reads all %d channels of the first result, not %d", variant, f.cout, g.cin);
    Y5_REQUIRE(g.cout > 0 && g.cout % 8 == 0 && g.y_ld % 8 == 0 && g.y_ld >= g.cout, "conv chain: cout %d / y_ld %d of the second convolution", g.cout, g.y_ld);
    Y5_REQUIRE(g.kpad % 8 == 0 && g.kpad >= (g.cin + 15) / 16 * 16, "conv chain: second weights need kpad >= %d", (g.cin + 15) / 16 * 16);
    Y5_REQUIRE(!g.scale || g.shift, "conv chain: scale without shift");
    Y5_REQUIRE(y5_aligned16(f.x) && y5_aligned16(f.w) && y5_aligned16(f.y) && y5_aligned16(f.res) && y5_aligned16(g.w) && y5_aligned16(g.y),
               "conv chain: pointers must be 16-byte aligned");
    for (const y5_conv_desc* d : {&f, &g}) {
        Y5_REQUIRE(d->act == Y5_ACT_NONE || (d->act == Y5_ACT_LEAKY && d->slope > 0.f && d->slope <= 1.f), "conv chain: bad activation");
        Y5_REQUIRE(y5_aligned16(d->bias) && y5_aligned16(d->scale) && y5_aligned16(d->shift), "conv chain: per-channel vectors must be 16-byte aligned");
    }
    if (variant == y5_conv_lw_count() + 1) {  // the weights-in-registers 128-channel kernel of conv_c128.hip
        int rc = Y5_OK;
        if (y5_try_conv_c128_chain(f, g, keep_first, s, &rc)) return rc;
        return y5_fail(Y5_EINVAL, "conv chain: the 128-channel variant takes 3x3 stride 1, 128 -> 128 -> 128 channels, width <= 40");
    }
    if (variant == y5_conv_lw_count() + 2 || (variant == 0 && part)) {  // the persistent 1x1 kernel of conv_p1x1.hip
        int rc = Y5_OK;
        if (y5_try_conv_p1x1_chain(f, g, keep_first, s, &rc)) return rc;
        return y5_fail(Y5_EINVAL, "conv chain: the persistent 1x1 variant takes 256 -> 256 -> (128 ->) 128 and 128 -> 128 -> (64 ->) 64 channels");
    }
    if (variant == y5_conv_lw_count() + 3) {  // the persistent 64-channel kernel of conv_c64.hip
        int rc = Y5_OK;
        if (y5_try_conv_c64_chain(f, g, keep_first, s, &rc)) return rc;
        return y5_fail(Y5_EINVAL, "conv chain: the last variant takes 3x3 stride 1, 64 -> 64 -> 64 channels, height %% 8 == 0, width %% 16 == 0");
    }
    const int need = f.cout > g.cout ? f.cout : g.cout;
    int which = variant;
    if (which == 0)
        for (int i = 1; i <= y5_conv_lw_count() && !which; ++i)
            if (kLw[i - 1].bn >= need) which = i;
    if (which == 0 || kLw[which - 1].bn < need) return y5_fail(Y5_EINVAL, "conv chain: no tile of variant %d holds %d channels", variant, need);
    const LwVariant& v = kLw[which - 1];
    Y5_REQUIRE(v.bn <= f.w_rows && v.bn <= g.w_rows, "conv chain: weights must be packed to at least %d rows", v.bn);
    ConvArgs a{};
    bool direct = false;
    if (!fill_args(f, a, &direct)) return y5_fail(Y5_EINVAL, "conv chain: tensors of 2 GiB or more");
    a.w2 = static_cast<const f16*>(g.w);
    a.bias2 = g.bias; a.scale2 = g.scale; a.shift2 = g.shift;
    a.y2 = static_cast<f16*>(g.y);
    a.w2_ld = g.kpad; a.y2_ld = g.y_ld; a.Cout2 = g.cout; a.keep_first = keep_first;
    a.slope2 = g.act == Y5_ACT_LEAKY ? g.slope : 1.f;
    return v.launch(a, direct, true, s);
}
