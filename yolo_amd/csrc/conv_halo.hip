// 3x3 stride-1 convolution with LDS-staged input halo tiles (gfx950).
//
// The im2col kernel in conv_igemm.hip re-gathers every input pixel nine times (once per tap).  Here a
// workgroup owns an 8x16 block of output pixels (x BN channels): for each chunk of 32 input channels it
// brings the 10x18 input halo into LDS ONCE by global->LDS DMA, and the nine taps read the same LDS image at
// shifted pixel offsets.  A step is one kernel row (three taps): its three 32-channel weight tiles are streamed
// per step (3-deep ring, two steps ahead); the next chunk's halo is fetched one chunk ahead into the second
// halo buffer.  All DMA is counted by hand (s_waitcnt vmcnt) with one barrier per step, i.e. three per chunk.
//
// LDS image: pixel records of 64 bytes (32 channels), halo rows at a pitch of 32 records so that the two
// 16-pixel rows an MFMA operand covers are 32 records apart; with the 16-byte chunk index XOR-ed with
// (record/4)%4 the ds_read_b128 fragment reads are bank-conflict free for every tap shift.
#include "y5_common.h"

namespace {

#include "conv_common.h"

#ifdef Y5_STAMPS
__device__ unsigned long long y5_stamph_buf[8 * 65536];
#define Y5_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define Y5_T(v)
#endif

constexpr int TH = 8, TW = 16;            // output pixels per workgroup
constexpr int HBM_ = TH * TW;             // 128 rows of the GEMM tile
constexpr int CK = 32;                    // input channels per chunk
constexpr int IH = TH + 2, IW = TW + 2;   // halo extent
constexpr int PITCH = 32;                 // records per halo row in LDS
constexpr int REC = CK * 2;               // bytes per pixel record
constexpr int HALO_BYTES = IH * PITCH * REC;
constexpr int HALO_DMA = IH * 2 / 4;      // DMA instructions per wave and chunk (two 16-record halves per row)
static_assert(IH * 2 % 4 == 0, "halo rows must split evenly over the four waves");

template <int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_halo3x3_kernel(const ConvArgs a) {
    static_assert(WM * WN == 4, "four waves per workgroup");
    constexpr int SB = 3;                            // weight ring depth
    constexpr int WTM = HBM_ / WM, WTN = BN / WN;    // wave tile: pixels x channels
    constexpr int RM = WTM / 32, RN = WTN / 32;
    static_assert(RM >= 1 && RN >= 1, "wave tile must hold at least one MFMA tile");
    constexpr int BT_BYTES = BN * REC;               // weight tile of one step
    constexpr bool B_HALF = BN * 4 < 256;            // BN == 32: waves 0,1 stage the weights
    constexpr int B_CH = (BN * 4 + 255) / 256;
    constexpr int EP = BN * 2 + 16;
    constexpr int RING0 = 2 * HALO_BYTES;
    constexpr int ST_BYTES = 3 * BT_BYTES;           // one ring stage: the three taps of a kernel row
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave % WM, wn = wave / WM;

    const int tile = xcd_remap(blockIdx.x, a.tiles_total);
    const int mtile = tile / a.tiles_n, nt = tile - mtile * a.tiles_n;
    const int n0 = nt * BN;
    constexpr int PRM0 = (RING0 + SB * ST_BYTES > HBM_ * EP) ? RING0 + SB * ST_BYTES : HBM_ * EP;
    float* prm = reinterpret_cast<float*>(smem + PRM0);
    const EpiParams epi = epi_fetch(a, n0, t, BN);
    const unsigned b = fdiv((unsigned)mtile, a.div_tpi);
    const unsigned rem = (unsigned)mtile - b * (unsigned)(a.tiles_x * a.tiles_y);
    const unsigned tyi = fdiv(rem, a.div_tx), txi = rem - tyi * (unsigned)a.tiles_x;
    const int oy0 = (int)tyi * TH, ox0 = (int)txi * TW;
    const f16* ximg = a.x + (size_t)b * a.H * a.W * a.x_ld;

    // ---- halo DMA state: this wave's instruction i covers LDS records [(y*PITCH + half*16), +16) with
    // j = wave*HALO_DMA + i, y = j/2, half = j%2; lane -> record (lane/4), 16-byte slot lane%4, which holds
    // channel chunk kc = (lane%4) ^ ((lane/16)%4) of the pixel (the record index is a multiple of 16 plus lane/4).
    const int kc = (lane & 3) ^ ((lane >> 4) & 3);
    const f16* hptr[HALO_DMA];
#pragma unroll
    for (int i = 0; i < HALO_DMA; ++i) {
        const int j = wave * HALO_DMA + i;
        const int y = j >> 1, xs = (j & 1) * 16 + (lane >> 2);
        const int iy = oy0 - 1 + y, ix = ox0 - 1 + xs;
        const bool ok = xs < IW && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        hptr[i] = ok ? ximg + ((size_t)iy * a.W + ix) * a.x_ld + kc * 8 : nullptr;
    }
    // ---- weight DMA state (as in the im2col kernel with 32-wide K-steps)
    const int row0 = t >> 2;
    const int kcb = (t & 3) ^ ((row0 >> 2) & 3);
    const bool bload = !B_HALF || (wave < 2);
    const f16* wrow[B_CH];
#pragma unroll
    for (int i = 0; i < B_CH; ++i) wrow[i] = a.w + (size_t)(n0 + (bload ? row0 : 0) + 64 * i) * a.w_ld + kcb * 8;
    const int lpsb = bload ? 3 * B_CH : 0;  // weight DMA instructions this wave issues per step
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

    auto issue_halo = [&](int ci, int buf) {
        const bool cvalid = ci * CK + kc * 8 < a.Cin;
#pragma unroll
        for (int i = 0; i < HALO_DMA; ++i) {
            const int j = wave * HALO_DMA + i;
            const void* src = (cvalid && hptr[i]) ? (const void*)(hptr[i] + ci * CK) : (const void*)y5_zero_page;
            dma16(src, lds0 + buf * HALO_BYTES + ((j >> 1) * PITCH + (j & 1) * 16) * REC);
        }
    };
    auto issue_b = [&](int ci, int r, int stage) {
        if (bload) {
            const bool cvalid = ci * CK + kcb * 8 < a.Cin;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int koff = (r * 3 + q) * a.Cin + ci * CK;
#pragma unroll
                for (int i = 0; i < B_CH; ++i)
                    dma16(cvalid ? (const void*)(wrow[i] + koff) : (const void*)y5_zero_page,
                          lds0 + RING0 + stage * ST_BYTES + q * BT_BYTES + i * 4096 + wave * 1024);
            }
        }
    };

    f32x16 acc[RN][RM];
#pragma unroll
    for (int j = 0; j < RN; ++j)
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;

    const int lr = lane & 31, lh = lane >> 5;
    const int tx = lr & 15, dy = lr >> 4;
    int arec[RM];  // LDS record of this lane's pixel for tap (0,0)
#pragma unroll
    for (int i = 0; i < RM; ++i) arec[i] = (wm * (WTM / 16) + 2 * i + dy) * PITCH + tx;
    const int swb = (lr >> 2) & 3;

#ifdef Y5_STAMPS
    unsigned long long st[3] = {0, 0, 0};
    const unsigned long long st_begin = __builtin_amdgcn_s_memtime();
#endif
    const int total = a.nchunks * 3;
    issue_halo(0, 0);
    issue_b(0, 0, 0);
    if (total > 1) issue_b(0, 1, 1);
    epi_store(prm, epi, t, BN);  // published by the barriers of the step loop
    const bool has_bias = a.bias != nullptr, has_scale = a.scale != nullptr;
    int it = 0;
    for (int ci = 0; ci < a.nchunks; ++ci) {
        const bool more = ci + 1 < a.nchunks;
        for (int r = 0; r < 3; ++r, ++it) {
            // younger than what this step reads: the next step's weights and, in rows 1 and 2, the next chunk's halo
            int allowed = (it + 1 < total) ? lpsb : 0;
            if (more && r >= 1) allowed += HALO_DMA;
            Y5_T(s0);
            wait_vm_and_barrier(allowed);
            Y5_T(s1);
            if (it + 2 < total) {
                const int nx = it + 2;
                const int nci = nx >= (ci + 1) * 3 ? ci + 1 : ci;
                issue_b(nci, nx - nci * 3, nx % SB);
            }
            if (r == 0 && more) issue_halo(ci + 1, (ci + 1) & 1);
            Y5_T(s2);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const unsigned char* Hb = smem + (ci & 1) * HALO_BYTES + (r * PITCH + q) * REC;
                const unsigned char* B = smem + RING0 + (it % SB) * ST_BYTES + q * BT_BYTES;
                const int swa = ((tx + q) >> 2) & 3;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    f16x8 xf[RM], wf[RN];
#pragma unroll
                    for (int i = 0; i < RM; ++i)
                        xf[i] = *reinterpret_cast<const f16x8*>(Hb + arec[i] * REC + (((2 * ks + lh) ^ swa) << 4));
#pragma unroll
                    for (int j = 0; j < RN; ++j)
                        wf[j] = *reinterpret_cast<const f16x8*>(B + (wn * WTN + j * 32 + lr) * REC + (((2 * ks + lh) ^ swb) << 4));
#pragma unroll
                    for (int j = 0; j < RN; ++j)
#pragma unroll
                        for (int i = 0; i < RM; ++i)
                            acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[j], xf[i], acc[j][i], 0, 0, 0);
                }
            }
#ifdef Y5_STAMPS
            const unsigned long long s3 = __builtin_amdgcn_s_memtime();
            st[0] += s1 - s0; st[1] += s2 - s1; st[2] += s3 - s2;
#endif
        }
    }
#ifdef Y5_STAMPS
    const unsigned long long st_loop_end = __builtin_amdgcn_s_memtime();
#endif
    wait_vm_and_barrier(0);

    // ---- epilogue (same staging as the im2col kernel; tile row p is pixel (p/16, p%16) of the block)
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
                    *reinterpret_cast<f16x4*>(smem + ml * EP + nl * 2) = post_op4_lds(v, nl, prm, BN, has_bias, has_scale, a.slope);
                }
            }
        }
    }
    __syncthreads();
    constexpr int CPR = BN / 8;
    constexpr int RPP = 256 / CPR;
    const int cc = t % CPR, rr = t / CPR;
    const int n = n0 + cc * 8;
    if (n < a.Cout) {
#pragma unroll
        for (int p = 0; p < HBM_ / RPP; ++p) {
            const int row = rr + p * RPP;
            const int oy = oy0 + (row >> 4), ox = ox0 + (row & 15);
            if (oy < a.H && ox < a.W) {
                const size_t m = ((size_t)b * a.H + oy) * a.W + ox;
                f16x8 v = *reinterpret_cast<const f16x8*>(smem + row * EP + cc * 16);
                if (a.res) v += *reinterpret_cast<const f16x8*>(a.res + m * a.res_ld + n);
                *reinterpret_cast<f16x8*>(a.y + m * a.y_ld + n) = v;
            }
        }
    }
#ifdef Y5_STAMPS
    if (lane == 0 && blockIdx.x * 4 + wave < 65536) {
        unsigned long long* o = y5_stamph_buf + (size_t)(blockIdx.x * 4 + wave) * 8;
        o[0] = st[0]; o[1] = st[1]; o[2] = st[2]; o[3] = st_loop_end - st_begin;
        o[4] = __builtin_amdgcn_s_memtime() - st_loop_end; o[5] = st_begin; o[6] = total;
    }
#endif
}

template <int BN, int WM, int WN>
int launch_halo(ConvArgs a, hipStream_t s) {
    a.tiles_n = (a.Cout + BN - 1) / BN;
    const long long total = (long long)a.tiles_x * a.tiles_y * (a.M / ((long long)a.H * a.W)) * a.tiles_n;
    if (total <= 0 || total > 0x7fffffffLL) return y5_fail(Y5_EINVAL, "conv: tile count %lld out of range", total);
    a.tiles_total = (int)total;
    constexpr int ring = 2 * HALO_BYTES + 3 * 3 * BN * REC, ep = HBM_ * (BN * 2 + 16);
    constexpr int smem = (ring > ep ? ring : ep) + 3 * BN * 4;
    static std::atomic<unsigned long long> attr_set{0};  // one bit per device: the attribute is per function and device
    const unsigned long long attr_set_bit = 1ull << y5_current_device();
    if (!(attr_set.load() & attr_set_bit)) {
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_halo3x3_kernel<BN, WM, WN>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set.fetch_or(attr_set_bit);
    }
    hipLaunchKernelGGL((conv_halo3x3_kernel<BN, WM, WN>), dim3((unsigned)total), dim3(256), smem, s, a);
    Y5_LAUNCH_CHECK("conv_halo3x3_kernel");
    return Y5_OK;
}

}  // namespace

#ifdef Y5_STAMPS
extern "C" int y5_debug_stampsh(unsigned long long* dst, int n_waves) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(y5_stamph_buf), (size_t)n_waves * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif

// Returns 1 when the halo-tile kernel handled the convolution, 0 when the caller should use the im2col kernel.
int y5_try_conv_halo(const y5_conv_desc& d, hipStream_t s, int force_bn, int* rc) {
    if (d.ksize != 3 || d.stride != 1) return 0;
    const int tiles_x = (d.wo + TW - 1) / TW, tiles_y = (d.ho + TH - 1) / TH;
    // partial tiles compute pixels that are never stored: by rule take this path only when few are wasted
    if (!force_bn && (long long)tiles_x * TW * tiles_y * TH * 4 > (long long)d.ho * d.wo * 5) return 0;
    ConvArgs a{};
    a.x = static_cast<const f16*>(d.x);
    a.w = static_cast<const f16*>(d.w);
    a.bias = d.bias; a.scale = d.scale; a.shift = d.shift;
    a.res = static_cast<const f16*>(d.res);
    a.y = static_cast<f16*>(d.y);
    a.M = (long long)d.batch * d.ho * d.wo;
    a.H = d.h; a.W = d.w_in; a.Cin = d.cin; a.x_ld = d.x_ld;
    a.Ho = d.ho; a.Wo = d.wo; a.Cout = d.cout; a.y_ld = d.y_ld; a.res_ld = d.res_ld;
    a.stride = 1; a.pad = 1; a.K = 9 * d.cin; a.w_ld = d.kpad;
    a.slope = d.act == Y5_ACT_LEAKY ? d.slope : 1.f;
    a.tiles_x = tiles_x; a.tiles_y = tiles_y; a.nchunks = (d.cin + CK - 1) / CK;
    a.div_tpi = make_fastdiv((unsigned)(tiles_x * tiles_y));
    a.div_tx = make_fastdiv((unsigned)tiles_x);
    auto waste = [&](int bn) { return ((d.cout + bn - 1) / bn) * bn - d.cout; };
    int bn = 128;
    if (waste(64) < waste(bn)) bn = 64;
    if (waste(32) < waste(bn) && d.cout <= 32) bn = 32;
    if (force_bn) bn = force_bn;
    if (bn == 128) *rc = launch_halo<128, 2, 2>(a, s);
    else if (bn == 64) *rc = launch_halo<64, 2, 2>(a, s);
    else *rc = launch_halo<32, 4, 1>(a, s);
    return 1;
}
