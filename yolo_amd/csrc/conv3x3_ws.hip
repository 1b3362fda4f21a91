// 3x3 convolution for narrow layers (32 or 64 input channels) on gfx950: weights resident, input staged once.
//
// The im2col kernel spends most of a K-step issuing global->LDS copies: every input pixel is fetched nine times (once
// per tap) and the weight tile again by every workgroup.  Here a persistent workgroup copies its BN x (9*CIN) weight
// slab into LDS once and then walks 8x16-pixel output tiles: the tile's input patch (10x18 pixels at stride 1, 17x33
// at stride 2, all CIN channels) goes to LDS once, through registers, one tile ahead of the MFMAs (two patch buffers),
// and the nine taps read it at shifted offsets.  One barrier per tile.
//
// Measured (cycle stamps, 64->64 at 80x80, batch 32): the products run at 71 % of the MFMA rate, but with the 75 KB
// slab only one workgroup fits a CU, so patch staging, epilogue and the slab load (a quarter of the kernel at 6 tiles
// per workgroup) are not overlapped by anything and the im2col kernel stays faster there; this kernel wins where a CU
// gets many tiles and two workgroups (32 channels at 160x160).  The engine's autotuner picks per layer.
//
// LDS images.  A pixel is a record of CIN*2 bytes whose 16-byte chunk c sits in slot c ^ f(record), f = (record/4)%4
// for 64-byte records and (record/2)%8 for 128-byte ones.  A wave multiplies 2 rows x 16 pixels; the record pitch
// between the rows a wave touches is a multiple of 16 records, so the 16 lanes of a ds_read_b128 group always read
// 16 records that are distinct modulo 16 and the reads are bank-conflict free for every tap.  At stride 2 the even and
// odd input columns are kept in separate planes so that the 16 pixels of a row are again consecutive records.
// The weight rows are padded by 16 bytes (an odd number of 16-byte slots) for the same reason.
#include "y5_common.h"

namespace {

struct WsArgs {
    const f16* x;
    const f16* w;
    const float *bias, *scale, *shift;
    const f16* res;
    f16* y;
    int B, H, W, x_ld, Ho, Wo, Cout, y_ld, res_ld, w_ld;
    int tiles_x, tiles_y, tiles_m, tiles_n;
    float slope;
};

constexpr int TH = 8, TW = 16;

#ifdef Y5_STAMPS
__device__ unsigned long long y5_stamp3_buf[8 * 65536];
#define Y5_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define Y5_T(v)
#endif

template <int CIN, int STRIDE>
struct Patch {
    static constexpr int NCH = CIN / 8;                       // 16-byte chunks per pixel record
    static constexpr int REC = CIN * 2;                       // bytes per record
    static constexpr int PR = STRIDE == 1 ? TH + 2 : 2 * TH + 1;  // patch rows
    static constexpr int PC = STRIDE == 1 ? TW + 2 : 2 * TW + 1;  // patch columns
    // stride 1: both buffers side by side in rows of 48 records (buffer b at column 24*b)
    // stride 2: rows of 40 records ([17 even columns][16 odd columns]), buffers 17*40 records apart
    static constexpr int PITCH = STRIDE == 1 ? 48 : 40;
    static constexpr int BUF = STRIDE == 1 ? 24 : PR * 40;    // records between the two buffers
    static constexpr int RECORDS = STRIDE == 1 ? PR * 48 : 2 * PR * 40;
    static constexpr int BYTES = RECORDS * REC;
    static constexpr int CHUNKS = PR * PC * NCH;              // 16-byte chunks staged per tile
    static constexpr int NSLOT = (CHUNKS + 255) / 256;
    __device__ static int record(int prow, int pcol) {
        if (STRIDE == 1) return prow * PITCH + pcol;
        return prow * PITCH + ((pcol & 1) ? 17 + (pcol >> 1) : (pcol >> 1));
    }
    __device__ static int swz(int rec) { return NCH == 4 ? (rec >> 2) & 3 : (rec >> 1) & 7; }
};

// Per-lane epilogue parameters of the 4-channel groups this lane owns (RN x 4 groups), kept in registers for the whole
// kernel: MODE bit 0 = bias, bit 1 = folded BatchNorm scale / shift.
template <int BN, int MODE>
struct LaneParams {
    f32x4 bias[MODE & 1 ? BN / 8 : 1], scale[MODE & 2 ? BN / 8 : 1], shift[MODE & 2 ? BN / 8 : 1];
};

template <int BN, int MODE>
__device__ __forceinline__ void finish_rows(const f32x16 (&acc)[BN / 32], const WsArgs& a, const LaneParams<BN, MODE>& prm,
                                            unsigned char* escr, int b, int oy0, int ox0, int n0, int lane) {
    constexpr int RN = BN / 32, EROW = BN * 2 + 16, CPW = BN / 8;
    const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int j = 0; j < RN; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int nl = j * 32 + 8 * g + 4 * lh;
            f32x4 v = {acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
            if (MODE & 1) v += prm.bias[j * 4 + g];
            f16x4 h = __builtin_convertvector(v, f16x4);
            if (MODE & 2) h = __builtin_convertvector(__builtin_convertvector(h, f32x4) * prm.scale[j * 4 + g] + prm.shift[j * 4 + g], f16x4);
            *reinterpret_cast<f16x4*>(escr + lr * EROW + nl * 2) = leaky4_f16(h, a.slope);
        }
    }
    constexpr int NP = 32 * CPW / 64;
    f16x8 rv[NP];
    size_t moff[NP];
    bool ok[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {  // residual loads first, all in flight together
        const int e = p * 64 + lane;
        const int px = e / CPW, cc = e - px * CPW;
        const int oy = oy0 + (px >> 4), ox = ox0 + (px & 15);
        ok[p] = oy < a.Ho && ox < a.Wo && n0 + cc * 8 < a.Cout;
        moff[p] = ((size_t)b * a.Ho + oy) * a.Wo + ox;
        if (a.res && ok[p]) rv[p] = *reinterpret_cast<const f16x8*>(a.res + moff[p] * a.res_ld + n0 + cc * 8);
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int e = p * 64 + lane;
        const int px = e / CPW, cc = e - px * CPW;
        if (ok[p]) {
            f16x8 v = *reinterpret_cast<const f16x8*>(escr + px * EROW + cc * 16);
            if (a.res) v += rv[p];
            *reinterpret_cast<f16x8*>(a.y + moff[p] * a.y_ld + n0 + cc * 8) = v;
        }
    }
}

template <int CIN, int BN, int STRIDE, int MODE, int DEPTH>
__global__ __launch_bounds__(256) void conv3x3_ws_kernel(const WsArgs a) {
    typedef Patch<CIN, STRIDE> P;
    constexpr int RN = BN / 32;
    constexpr int WPITCH = 9 * CIN * 2 + 16;        // bytes per weight row in LDS
    constexpr int EROW = BN * 2 + 16;
    constexpr int W0 = 0, P0 = BN * WPITCH, E0 = P0 + P::BYTES;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

#ifdef Y5_STAMPS
    const unsigned long long st_begin = __builtin_amdgcn_s_memtime();
#endif
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int nt = blockIdx.x % a.tiles_n;
    const int slot0 = blockIdx.x / a.tiles_n, nslots = gridDim.x / a.tiles_n;
    const int n0 = nt * BN;

    // ---- weight slab, once per workgroup: all loads first, then the LDS stores
    {
        constexpr int WCH = BN * 9 * (CIN / 8), WPT = (WCH + 255) / 256;
        uint4 wv[WPT];
#pragma unroll
        for (int k = 0; k < WPT; ++k) {
            const int i = k * 256 + t;
            const int n = i / (9 * (CIN / 8)), kc = i - n * (9 * (CIN / 8));
            if (i < WCH) wv[k] = *reinterpret_cast<const uint4*>(a.w + (size_t)(n0 + n) * a.w_ld + kc * 8);
        }
#pragma unroll
        for (int k = 0; k < WPT; ++k) {
            const int i = k * 256 + t;
            const int n = i / (9 * (CIN / 8)), kc = i - n * (9 * (CIN / 8));
            if (i < WCH) *reinterpret_cast<uint4*>(smem + W0 + n * WPITCH + kc * 16) = wv[k];
        }
    }
    // ---- this lane's epilogue parameters (channels beyond Cout exist in the padded vectors)
    LaneParams<BN, MODE> prm;
#pragma unroll
    for (int j = 0; j < RN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0 + j * 32 + 8 * g + 4 * lh;
            if (MODE & 1) prm.bias[j * 4 + g] = n < a.Cout ? *reinterpret_cast<const f32x4*>(a.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            if (MODE & 2) {
                prm.scale[j * 4 + g] = n < a.Cout ? *reinterpret_cast<const f32x4*>(a.scale + n) : f32x4{1.f, 1.f, 1.f, 1.f};
                prm.shift[j * 4 + g] = n < a.Cout ? *reinterpret_cast<const f32x4*>(a.shift + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }

    // ---- patch staging state: thread t owns chunks i*256 + t of the patch; where they come from (relative to the
    // patch origin) and where they go in LDS never changes
    int goff[P::NSLOT];   // element offset from the patch's first pixel
    int meta[P::NSLOT];   // LDS byte offset in buffer 0 | pcol << 18 | prow << 25, or -1
#pragma unroll
    for (int i = 0; i < P::NSLOT; ++i) {
        const int id = i * 256 + t;
        const int c = id % P::NCH, pix = id / P::NCH;
        const int prow = pix / P::PC, pcol = pix - prow * P::PC;
        const int rec = P::record(prow, pcol);
        goff[i] = (prow * a.W + pcol) * a.x_ld + c * 8;
        meta[i] = id < P::CHUNKS ? ((rec * P::REC + ((c ^ P::swz(rec)) << 4)) | pcol << 18 | prow << 25) : -1;
    }
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    // Input patches are fetched DEPTH tiles ahead into registers (set k holds a tile whose sequence number is k modulo
    // DEPTH): with one workgroup per CU a single patch in flight leaves the tile period equal to the memory latency.
    uint4 pre[DEPTH][P::NSLOT];
    auto fetch = [&](int tile, uint4 (&pre)[P::NSLOT]) {
        const int b = tile / tiles_per_img, rem = tile - b * tiles_per_img;
        const int iy0 = (rem / a.tiles_x) * TH * STRIDE - 1, ix0 = (rem % a.tiles_x) * TW * STRIDE - 1;
        const f16* origin = a.x + (((ptrdiff_t)b * a.H + iy0) * a.W + ix0) * a.x_ld;
        const bool interior = iy0 >= 0 && iy0 + P::PR <= a.H && ix0 >= 0 && ix0 + P::PC <= a.W;
        if (interior) {
#pragma unroll
            for (int i = 0; i < P::NSLOT; ++i)
                if (meta[i] >= 0) pre[i] = *reinterpret_cast<const uint4*>(origin + goff[i]);
        } else {
#pragma unroll
            for (int i = 0; i < P::NSLOT; ++i) {
                const int iy = iy0 + (meta[i] >> 25), ix = ix0 + ((meta[i] >> 18) & 127);
                const bool ok = meta[i] >= 0 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                const uint4 v = *reinterpret_cast<const uint4*>(ok ? origin + goff[i] : a.x);
                pre[i] = ok ? v : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto stage = [&](int buf, const uint4 (&pre)[P::NSLOT]) {
#pragma unroll
        for (int i = 0; i < P::NSLOT; ++i)
            if (meta[i] >= 0) *reinterpret_cast<uint4*>(smem + P0 + buf * P::BUF * P::REC + (meta[i] & 0x3ffff)) = pre[i];
    };

    const int my_tiles = slot0 < a.tiles_m ? (a.tiles_m - slot0 + nslots - 1) / nslots : 0;
    if (my_tiles > 0) {
        fetch(slot0, pre[0]);
        stage(0, pre[0]);
#pragma unroll
        for (int k = 1; k <= DEPTH; ++k)
            if (k < my_tiles) fetch(slot0 + k * nslots, pre[k % DEPTH]);
    }
    __syncthreads();  // weights, parameters and patch 0 are in place

    // this lane's pixel: row dy of the wave's two, column tx
    const int dy = lr >> 4, tx = lr & 15;
    const int rec0 = STRIDE == 1 ? (2 * wave + dy) * P::PITCH + tx : (2 * (2 * wave + dy)) * P::PITCH + tx;
    unsigned char* escr = smem + E0 + wave * 32 * EROW;
    int cur = 0;

#ifdef Y5_STAMPS
    unsigned long long st[4] = {0, 0, 0, 0};
    const unsigned long long st_loop = __builtin_amdgcn_s_memtime();
#endif
    for (int tb = 0; tb < my_tiles; tb += DEPTH) {
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {  // unrolled so that the register set of a tile is a compile-time choice
        const int ti = tb + u;
        if (ti >= my_tiles) break;
        const int tile = slot0 + ti * nslots;
        const int b = tile / tiles_per_img, rem = tile - b * tiles_per_img;
        const int oy0 = (rem / a.tiles_x) * TH, ox0 = (rem % a.tiles_x) * TW;
        const unsigned char* pb = smem + P0 + cur * P::BUF * P::REC;

        f32x16 acc[RN];
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        Y5_T(s0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int r = tap / 3, q = tap - 3 * r;
            // record of this lane's pixel for the tap: a compile-time shift of rec0
            const int shift = STRIDE == 1 ? r * P::PITCH + q : r * P::PITCH + (q == 1 ? 17 : q == 2 ? 1 : 0);
            const int rec = rec0 + shift;
            const int sw = P::swz(rec);  // buffer-local record index on both sides; the buffers are a multiple of 4 records apart
            const unsigned char* prec = pb + rec * P::REC;
#pragma unroll
            for (int ks = 0; ks < CIN / 16; ++ks) {
                const f16x8 xf = *reinterpret_cast<const f16x8*>(prec + (((2 * ks + lh) ^ sw) << 4));
#pragma unroll
                for (int j = 0; j < RN; ++j) {
                    const f16x8 wf = *reinterpret_cast<const f16x8*>(smem + W0 + (j * 32 + lr) * WPITCH + (tap * (CIN / 8) + 2 * ks + lh) * 16);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, xf, acc[j], 0, 0, 0);
                }
            }
        }

        Y5_T(s1);
        if (ti + 1 < my_tiles) {
            // buffer cur^1 was last read by the previous tile's products, which every wave finished before the last barrier
            stage(cur ^ 1, pre[(u + 1) % DEPTH]);
            if (ti + 1 + DEPTH < my_tiles) fetch(tile + (1 + DEPTH) * nslots, pre[(u + 1) % DEPTH]);
        }
        Y5_T(s2);
        const int oyw = oy0 + 2 * wave;
        finish_rows<BN, MODE>(acc, a, prm, escr, b, oyw, ox0, n0, lane);
        Y5_T(s3);
        __syncthreads();
#ifdef Y5_STAMPS
        const unsigned long long s4 = __builtin_amdgcn_s_memtime();
        st[0] += s1 - s0; st[1] += s2 - s1; st[2] += s3 - s2; st[3] += s4 - s3;
#endif
        cur ^= 1;
    }
    }
#ifdef Y5_STAMPS
    if (lane == 0 && blockIdx.x * 4 + wave < 65536) {
        unsigned long long* o = y5_stamp3_buf + (size_t)(blockIdx.x * 4 + wave) * 8;
        o[0] = st[0]; o[1] = st[1]; o[2] = st[2]; o[3] = st[3]; o[4] = __builtin_amdgcn_s_memtime() - st_begin; o[5] = st_loop - st_begin; o[6] = my_tiles;
    }
#endif
}

template <int CIN, int BN, int STRIDE, int MODE>
int launch_ws3m(WsArgs a, hipStream_t s) {
    typedef Patch<CIN, STRIDE> P;
    constexpr size_t lds = (size_t)BN * (9 * CIN * 2 + 16) + P::BYTES + 4 * 32 * (BN * 2 + 16);
    static_assert(lds <= 160 * 1024, "configuration does not fit in LDS");
    // a configuration that leaves room for only one workgroup per CU has registers to spare: fetch two tiles ahead
    constexpr int DEPTH = lds > 80 * 1024 ? 2 : 1;
    auto kern = conv3x3_ws_kernel<CIN, BN, STRIDE, MODE, DEPTH>;
    static std::atomic<unsigned long long> attr_set{0};  // one bit per device: the attribute is per function and device
    const unsigned long long attr_set_bit = 1ull << y5_current_device();
    if (!(attr_set.load() & attr_set_bit)) {
        Y5_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set.fetch_or(attr_set_bit);
    }
    a.tiles_n = (a.Cout + BN - 1) / BN;
    const int per_cu = (int)((160 * 1024) / lds) > 2 ? 2 : (int)((160 * 1024) / lds);
    long long slots = (256LL * per_cu) / a.tiles_n;
    if (slots < 1) slots = 1;
    if (slots > a.tiles_m) slots = a.tiles_m;
    hipLaunchKernelGGL(kern, dim3((unsigned)(slots * a.tiles_n)), dim3(256), lds, s, a);
    Y5_LAUNCH_CHECK("conv3x3_ws_kernel");
    return Y5_OK;
}

template <int CIN, int BN, int STRIDE>
int launch_ws3(const WsArgs& a, hipStream_t s) {
    switch ((a.bias ? 1 : 0) | (a.scale ? 2 : 0)) {
        case 0: return launch_ws3m<CIN, BN, STRIDE, 0>(a, s);
        case 1: return launch_ws3m<CIN, BN, STRIDE, 1>(a, s);
        case 2: return launch_ws3m<CIN, BN, STRIDE, 2>(a, s);
        default: return launch_ws3m<CIN, BN, STRIDE, 3>(a, s);
    }
}

}  // namespace

#ifdef Y5_STAMPS
extern "C" int y5_debug_stamps3(unsigned long long* dst, int n_waves) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(y5_stamp3_buf), (size_t)n_waves * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif

// which = 1: BN = 64, 2: BN = 32.  Returns 1 and sets *rc when this kernel took the convolution, 0 when it does not apply
// (not 3x3, input channels other than 32 / 64, stride 2 with 64 channels, or a tensor too large for 32-bit offsets).
int y5_try_conv3x3_ws(const y5_conv_desc& d, hipStream_t s, int which, int* rc) {
    if (d.ksize != 3 || d.pad != 1 || (d.stride != 1 && d.stride != 2)) return 0;
    if (d.cin != 32 && d.cin != 64) return 0;
    if (d.stride == 2 && d.cin != 32) return 0;
    if ((long long)d.h * d.w_in * d.x_ld >= (1LL << 31)) return 0;
    WsArgs a{};
    a.x = static_cast<const f16*>(d.x);
    a.w = static_cast<const f16*>(d.w);
    a.bias = d.bias; a.scale = d.scale; a.shift = d.shift;
    a.res = static_cast<const f16*>(d.res);
    a.y = static_cast<f16*>(d.y);
    a.B = d.batch; a.H = d.h; a.W = d.w_in; a.x_ld = d.x_ld; a.Ho = d.ho; a.Wo = d.wo;
    a.Cout = d.cout; a.y_ld = d.y_ld; a.res_ld = d.res_ld; a.w_ld = d.kpad;
    a.tiles_x = (d.wo + TW - 1) / TW; a.tiles_y = (d.ho + TH - 1) / TH;
    const long long tm = (long long)a.tiles_x * a.tiles_y * d.batch;
    if (tm > 0x7fffffffLL) return 0;
    a.tiles_m = (int)tm;
    a.slope = d.act == Y5_ACT_LEAKY ? d.slope : 1.f;
    const int bn = which == 2 ? 32 : 64;
#define Y5_W3(CIN, BN, ST) *rc = launch_ws3<CIN, BN, ST>(a, s); return 1
    if (d.stride == 1) {
        if (d.cin == 32) { if (bn == 64) { Y5_W3(32, 64, 1); } Y5_W3(32, 32, 1); }
        if (bn == 64) { Y5_W3(64, 64, 1); }
        Y5_W3(64, 32, 1);
    }
    if (bn == 64) { Y5_W3(32, 64, 2); }
    Y5_W3(32, 32, 2);
#undef Y5_W3
}
