// Implicit-GEMM convolution on MFMA for gfx950: NHWC fp16 activations, fp32 accumulation,
// fused bias / folded-BatchNorm / LeakyReLU / residual epilogue, output into a channel slice.
//
// GEMM view:  D[n][m] = sum_k Wt[n][k] * X[m][k]
//   m = (b, oy, ox) output pixel, n = output channel, k = (r*ks + q)*cin + c.
// One 256-thread workgroup (4 waves) owns a BM-pixel x BN-channel tile and walks K in steps of 32:
// the X tile is gathered (im2col on the fly, zero for padding) and the Wt tile copied with 16-byte
// global loads into registers, written to a double-buffered LDS image with a 16-byte XOR swizzle,
// and consumed by v_mfma_f32_32x32x16_f16 with the weights as the row operand, so that every lane
// ends up holding 4 consecutive channels of one pixel per accumulator group.  The epilogue applies
// the per-channel post-ops in fp32 with the reference's fp16 rounding points, stages the tile in
// LDS and writes whole 16-byte channel vectors, adding the residual there.
#include "y5_common.h"

namespace {

struct ConvArgs {
    const f16* x;
    const f16* w;
    const float* bias;
    const float* scale;
    const float* shift;
    const f16* res;
    f16* y;
    long long M;  // batch*ho*wo
    int H, W, Cin, x_ld;
    int Ho, Wo, Cout, y_ld, res_ld;
    int stride, pad, K, kpad;
    int tiles_n, tiles_total;
    int act;
    float slope;
};

constexpr int BK = 32;          // K elements per step (four 16-byte chunks per tile row)
constexpr int ROW_BYTES = 64;   // LDS bytes per tile row and step

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // Workgroups are dealt round-robin over the 8 XCDs; give each XCD one contiguous run of tiles
    // so neighbouring tiles (which share input rows and weights) share an L2.  Bijective for any nwg.
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + (bid >> 3);
}

__device__ __forceinline__ f16 post_op(float v, int n, const ConvArgs& a) {
    if (a.bias) v += a.bias[n];
    f16 h = (f16)v;
    if (a.scale) h = (f16)((float)h * a.scale[n] + a.shift[n]);
    if (a.act == Y5_ACT_LEAKY) {
        const float f = (float)h;
        h = (f16)(f > 0.f ? f : f * a.slope);
    }
    return h;
}

template <int BM, int BN, int WM, int WN, int KS>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs a) {
    static_assert(WM * WN == 4, "four waves per workgroup");
    constexpr int WTM = BM / WM, WTN = BN / WN;  // wave tile: pixels x channels
    constexpr int RM = WTM / 32, RN = WTN / 32;  // 32x32 MFMA tiles per wave
    static_assert(RM >= 1 && RN >= 1, "wave tile must hold at least one MFMA tile");
    constexpr int A_CH = BM * 4 / 256;               // 16-byte activation chunks per thread and step
    constexpr int B_CH = (BN * 4 + 255) / 256;       // 16-byte weight chunks per thread and step
    constexpr int STAGE_BYTES = (BM + BN) * ROW_BYTES;
    constexpr int EP = BN * 2 + 16;                  // epilogue row pitch in bytes
    constexpr int SMEM = (2 * STAGE_BYTES > BM * EP) ? 2 * STAGE_BYTES : BM * EP;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int wm = wave % WM, wn = wave / WM;

    const int tile = xcd_remap(blockIdx.x, a.tiles_total);
    const int mt = tile / a.tiles_n, nt = tile - mt * a.tiles_n;
    const long long m0 = (long long)mt * BM;
    const int n0 = nt * BN;

    // ---- per-thread gather state: each thread owns chunk column kc of A_CH activation rows ----
    const int kc = t & 3;
    const int row0 = t >> 2;  // 0..63
    int hi0[A_CH], wi0[A_CH];
    long long pix0[A_CH];  // pixel index of (b, 0, 0)
    bool mvalid[A_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const long long m = m0 + row0 + 64 * i;
        mvalid[i] = m < a.M;
        const long long mm = mvalid[i] ? m : 0;
        const int hw = a.Ho * a.Wo;
        const long long b = mm / hw;
        const int rem = (int)(mm - b * hw);
        const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
        hi0[i] = oy * a.stride - a.pad;
        wi0[i] = ox * a.stride - a.pad;
        pix0[i] = b * (long long)a.H * a.W;
    }
    const f16* wrow[B_CH];
#pragma unroll
    for (int i = 0; i < B_CH; ++i) wrow[i] = a.w + (size_t)(n0 + row0 + 64 * i) * a.kpad + kc * 8;
    const bool bload = (BN * 4 >= 256) || (t < BN * 4);

    uint4 areg[A_CH], breg[B_CH];
    auto load_tiles = [&](int k0) {
        const int k = k0 + kc * 8;
        int r = 0, q = 0, c = k;
        if (KS > 1) {
            const int tap = k / a.Cin;
            c = k - tap * a.Cin;
            r = tap / KS;
            q = tap - r * KS;
        }
        const bool kvalid = k < a.K;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const int hi = hi0[i] + r, wi = wi0[i] + q;
            const bool ok = kvalid && mvalid[i] && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok) v = *reinterpret_cast<const uint4*>(a.x + (size_t)(pix0[i] + (long long)hi * a.W + wi) * a.x_ld + c);
            areg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_CH; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (bload) v = *reinterpret_cast<const uint4*>(wrow[i] + k0);
            breg[i] = v;
        }
    };
    auto store_tiles = [&](int stage) {
        unsigned char* A = smem + stage * STAGE_BYTES;
        unsigned char* B = A + BM * ROW_BYTES;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const int row = row0 + 64 * i;
            *reinterpret_cast<uint4*>(A + row * ROW_BYTES + ((kc ^ ((row >> 2) & 3)) << 4)) = areg[i];
        }
#pragma unroll
        for (int i = 0; i < B_CH; ++i) {
            const int row = row0 + 64 * i;
            if (bload) *reinterpret_cast<uint4*>(B + row * ROW_BYTES + ((kc ^ ((row >> 2) & 3)) << 4)) = breg[i];
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
    const int sw = (lr >> 2) & 3;  // tile rows handled by a wave start at multiples of 32
    const int nk = a.kpad / BK;

    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tiles((kt + 1) * BK);
        const unsigned char* A = smem + cur * STAGE_BYTES;
        const unsigned char* B = A + BM * ROW_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
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
        if (kt + 1 < nk) store_tiles(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: post-ops in registers -> fp16 tile in LDS -> 16-byte stores (+ residual) ----
    // accumulator element e of tile (j,i): channel = 8*(e>>2) + 4*lh + (e&3) within the 32-channel
    // MFMA tile, pixel = lr.
#pragma unroll
    for (int j = 0; j < RN; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int nl = wn * WTN + j * 32 + 8 * g + 4 * lh;
            const int n = n0 + nl;
            if (n < a.Cout) {
#pragma unroll
                for (int i = 0; i < RM; ++i) {
                    f16x4 h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = post_op(acc[j][i][4 * g + e], n + e, a);
                    const int ml = wm * WTM + i * 32 + lr;
                    *reinterpret_cast<f16x4*>(smem + ml * EP + nl * 2) = h;
                }
            }
        }
    }
    __syncthreads();
    constexpr int CPR = BN / 8;          // 16-byte chunks per tile row
    constexpr int RPP = 256 / CPR;       // rows per pass
    const int cc = t % CPR, rr = t / CPR;
    const int n = n0 + cc * 8;
    if (n < a.Cout) {
#pragma unroll
        for (int p = 0; p < BM / RPP; ++p) {
            const int row = rr + p * RPP;
            const long long m = m0 + row;
            if (m < a.M) {
                f16x8 v = *reinterpret_cast<const f16x8*>(smem + row * EP + cc * 16);
                if (a.res) v += *reinterpret_cast<const f16x8*>(a.res + (size_t)m * a.res_ld + n);
                *reinterpret_cast<f16x8*>(a.y + (size_t)m * a.y_ld + n) = v;
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
int launch_tile(const ConvArgs& a0, int ksize, hipStream_t s) {
    ConvArgs a = a0;
    const long long tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.Cout + BN - 1) / BN;
    const long long total = tiles_m * a.tiles_n;
    if (total <= 0 || total > 0x7fffffffLL) return y5_fail(Y5_EINVAL, "conv: tile count %lld out of range", total);
    a.tiles_total = (int)total;
    if (ksize == 1)
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, 1>), dim3((unsigned)total), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, 3>), dim3((unsigned)total), dim3(256), 0, s, a);
    Y5_LAUNCH_CHECK("conv_igemm_kernel");
    return Y5_OK;
}

}  // namespace

int y5_launch_conv(const y5_conv_desc& d, hipStream_t s) {
    Y5_REQUIRE(d.x && d.w && d.y, "conv: null tensor pointer");
    Y5_REQUIRE(d.ksize == 1 || d.ksize == 3, "conv: ksize %d not in {1,3}", d.ksize);
    Y5_REQUIRE(d.stride == 1 || d.stride == 2, "conv: stride %d not in {1,2}", d.stride);
    Y5_REQUIRE(d.pad == d.ksize / 2, "conv: pad %d != ksize/2", d.pad);
    Y5_REQUIRE(d.batch > 0 && d.h > 0 && d.w_in > 0 && d.ho > 0 && d.wo > 0, "conv: empty tensor");
    Y5_REQUIRE(d.ho == (d.h + 2 * d.pad - d.ksize) / d.stride + 1 && d.wo == (d.w_in + 2 * d.pad - d.ksize) / d.stride + 1,
               "conv: output %dx%d inconsistent with input %dx%d k%d s%d", d.ho, d.wo, d.h, d.w_in, d.ksize, d.stride);
    Y5_REQUIRE(d.cin > 0 && d.cin % 8 == 0 && d.x_ld % 8 == 0 && d.x_ld >= d.cin, "conv: cin %d / x_ld %d must be multiples of 8", d.cin, d.x_ld);
    Y5_REQUIRE(d.cout > 0 && d.cout % 8 == 0 && d.y_ld % 8 == 0 && d.y_ld >= d.cout, "conv: cout %d / y_ld %d must be multiples of 8", d.cout, d.y_ld);
    Y5_REQUIRE(d.kpad % 32 == 0 && d.kpad >= d.ksize * d.ksize * d.cin, "conv: kpad %d too small or not a multiple of 32", d.kpad);
    Y5_REQUIRE(d.w_rows % 128 == 0 && d.w_rows >= d.cout, "conv: w_rows %d must be a multiple of 128 >= cout", d.w_rows);
    Y5_REQUIRE(!d.scale || d.shift, "conv: scale without shift");
    Y5_REQUIRE(!d.res || (d.res_ld % 8 == 0 && d.res_ld >= d.cout), "conv: bad residual pitch %d", d.res_ld);
    Y5_REQUIRE(y5_aligned16(d.x) && y5_aligned16(d.w) && y5_aligned16(d.y) && y5_aligned16(d.res), "conv: pointers must be 16-byte aligned");
    Y5_REQUIRE(d.act == Y5_ACT_NONE || d.act == Y5_ACT_LEAKY, "conv: unknown activation %d", d.act);

    ConvArgs a;
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
    a.stride = d.stride; a.pad = d.pad; a.K = d.ksize * d.ksize * d.cin; a.kpad = d.kpad;
    a.tiles_n = 0; a.tiles_total = 0;
    a.act = d.act; a.slope = d.slope;

    // Channel-tile width: the widest of {128, 64, 32} that wastes the fewest padded channels.
    auto waste = [&](int bn) { return ((d.cout + bn - 1) / bn) * bn - d.cout; };
    int bn = 128;
    if (waste(64) < waste(bn)) bn = 64;
    if (waste(32) < waste(bn) && d.cout <= 32) bn = 32;
    if (bn == 128) return launch_tile<128, 128, 2, 2>(a, d.ksize, s);
    if (bn == 64) return launch_tile<128, 64, 2, 2>(a, d.ksize, s);
    return launch_tile<128, 32, 4, 1>(a, d.ksize, s);
}
