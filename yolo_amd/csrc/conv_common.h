// Pieces shared by the convolution kernels (included inside an anonymous namespace by each translation unit).
#pragma once

struct FastDiv {  // n / d for 0 <= n < 2^31 by multiply-high and shift
    unsigned mul, shr, d;
};

inline FastDiv make_fastdiv(unsigned d) {
    FastDiv f{0u, 0u, d};
    if (d > 1) {
        unsigned lg = 0;
        while ((1ull << lg) < d) ++lg;
        const unsigned p = 31 + lg;
        f.mul = (unsigned)(((1ull << p) + d - 1) / d);
        f.shr = p - 32;
    }
    return f;
}

__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) { return f.d == 1 ? n : (__umulhi(n, f.mul) >> f.shr); }

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
    int stride, pad, K;
    int w_ld;     // weight row pitch in elements
    int nk;       // K-steps to walk
    int tiles_n, tiles_total;
    float slope;  // 1.0 = no activation: leaky(v) = max(v, slope*v) for 0 < slope <= 1
    FastDiv div_hw, div_wo, div_cin;
    // halo-tile kernel only: output tiles per image and channel chunks
    int tiles_x, tiles_y, nchunks;
    FastDiv div_tpi, div_tx;
    // loader-wave kernel, buffer addressing: bytes the x / w descriptors span
    int x_bytes, w_bytes;
    // loader-wave kernel, chained 1x1 convolution on the finished tile (conv_lw.hip, TAIL)
    const f16* w2;
    const float *bias2, *scale2, *shift2;
    f16* y2;
    int w2_ld, y2_ld, Cout2, keep_first;
    float slope2;
    // 1x1 convolution of a concatenation (conv_igemm.hip, CAT2): input channels [0, Cin) are x, a tensor of half the
    // output's height and width read at (oy/2, ox/2) (nearest 2x upsampling); channels [Cin, K) are x2 at (oy, ox)
    const f16* x2;
    int x2_ld, x2_bytes;
    // per-channel sums of the stored outputs (conv_igemm.hip epilogue): stats[r][0..stats_c) += sum, [stats_c..2 stats_c) +=
    // sum of squares, r = M-tile index % stats_rows, channel n at column stats_off + n; nullptr = off
    float* stats;
    int stats_rows, stats_c, stats_off;
    // linear-halo 3x3 kernel (conv_lin3.hip): bytes per staged pixel (channels + one 16-byte slot of padding), staged
    // pixels, 1 KiB DMA pieces of the halo, input pixels of the whole batch
    int lin_pitch, lin_hpx, lin_pieces;
    long long lin_min;
    FastDiv div_slots;
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // Workgroups are dealt round-robin over the 8 XCDs; give each XCD one contiguous run of tiles
    // so neighbouring tiles (which share input rows and weights) share an L2.  Bijective for any nwg.
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + (bid >> 3);
}

// 64 bytes of zeros: the source of every out-of-image / out-of-range 16-byte chunk, so that the
// LDS-DMA below can always write a full tile without branching.
static __device__ __attribute__((aligned(64))) unsigned int y5_zero_page[16];

// One 16-byte-per-lane global -> LDS DMA (global_load_lds_dwordx4): the wave writes 1 KiB starting at the
// wave-uniform LDS byte address lds_base, lane i at lds_base + 16*i, from each lane's own source address.
// Issued from inline asm so that the compiler neither counts it nor drains it before later LDS reads;
// completion is tracked by hand with s_waitcnt vmcnt(N).
__device__ __forceinline__ void dma16(const void* src, unsigned lds_base) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(src), "s"(lds_base)
        : "memory");
}

// The same DMA through a buffer descriptor: source = descriptor base + voff (bytes, per lane); a lane whose offset fails
// the descriptor's range check (voff >= num_records) receives zeros, which replaces a zero page for padding.  Measured
// beside MFMA work on the same SIMD, this form costs the SIMD far less issue time than the 64-bit-address form above.
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned Y5_BUF_OOB = 0x80000000u;  // >= any num_records used here (spans are kept below 2^31 bytes)
__device__ __forceinline__ i32x4 make_buf(const void* p, int bytes) {
    const unsigned long long v = (unsigned long long)p;
    const i32x4 r = {(int)(unsigned)v, (int)((unsigned)(v >> 32) & 0xffffu), bytes, 0x00020000};  // raw buffer, stride 0
    return r;
}
__device__ __forceinline__ void dma16_buf(i32x4 rsrc, unsigned voff, unsigned lds_base) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(rsrc), "s"(lds_base)
        : "memory");
}

// buffer_load_dwordx4 ... lds: the wave writes 1 KiB at the wave-uniform LDS address dst, lane i at dst + 16*i, from
// descriptor base + voff (per lane) + soff (wave-uniform); a lane whose offset fails the descriptor's range check gets
// zeros.  The descriptor type and builtins exist in the device pass only.
typedef __attribute__((address_space(3))) unsigned char* lds_ptr;
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t BufRsrc;
__device__ __forceinline__ BufRsrc make_rsrc(const void* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void buf_dma16(BufRsrc r, lds_ptr dst, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, voff, soff, 0, 0);
}
#else
struct BufRsrc {};
__device__ inline BufRsrc make_rsrc(const void*, int) { return BufRsrc{}; }
__device__ inline void buf_dma16(BufRsrc, lds_ptr, unsigned, int) {}
#endif

// Wait until at most n of this wave's vector-memory operations are outstanding, then meet the workgroup.
__device__ __forceinline__ void wait_vm_and_barrier(int n) {
#define Y5_WAIT_CASE(N) \
    case N: asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    switch (n) {
        Y5_WAIT_CASE(1) Y5_WAIT_CASE(2) Y5_WAIT_CASE(3) Y5_WAIT_CASE(4) Y5_WAIT_CASE(5) Y5_WAIT_CASE(6)
        Y5_WAIT_CASE(7) Y5_WAIT_CASE(8) Y5_WAIT_CASE(9) Y5_WAIT_CASE(10) Y5_WAIT_CASE(11) Y5_WAIT_CASE(12) Y5_WAIT_CASE(16)
        default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    }
#undef Y5_WAIT_CASE
}

// Per-channel epilogue parameters of a workgroup's BN channels, staged in LDS as [bias | scale | shift][BN] floats at
// kernel start: read from global memory in the epilogue they would cost a dependent memory round trip per workgroup
// just when nothing else is left to overlap it.  fetch -> (issue the prologue DMA) -> store.
struct EpiParams {
    float b, s, h;
};
__device__ __forceinline__ EpiParams epi_fetch(const ConvArgs& a, int n0, int t, int bn) {
    EpiParams p = {0.f, 1.f, 0.f};
    const int n = n0 + t;
    if (t < bn && n < a.Cout) {
        if (a.bias) p.b = a.bias[n];
        if (a.scale) {
            p.s = a.scale[n];
            p.h = a.shift[n];
        }
    }
    return p;
}
__device__ __forceinline__ void epi_store(float* prm, const EpiParams& p, int t, int bn) {
    if (t < bn) {
        prm[t] = p.b;
        prm[bn + t] = p.s;
        prm[2 * bn + t] = p.h;
    }
}
// Post-ops on four consecutive channels nl..nl+3 (workgroup-local, nl % 4 == 0) of one pixel, parameters from LDS.
__device__ __forceinline__ f16x4 post_op4_lds_f16(f32x4 v, int nl, const float* prm, int bn, bool has_bias, bool has_scale, float slope) {
    if (has_bias) v += *reinterpret_cast<const f32x4*>(prm + nl);
    f16x4 h = __builtin_convertvector(v, f16x4);
    if (has_scale)
        h = __builtin_convertvector(__builtin_convertvector(h, f32x4) * *reinterpret_cast<const f32x4*>(prm + bn + nl) +
                                        *reinterpret_cast<const f32x4*>(prm + 2 * bn + nl),
                                    f16x4);
    return leaky4_f16(h, slope);
}
template <typename ET = f16>
__device__ __forceinline__ typename Elem<ET>::v4 post_op4_lds(f32x4 v, int nl, const float* prm, int bn, bool has_bias, bool has_scale, float slope) {
    if constexpr (sizeof(typename Elem<ET>::v4) == sizeof(f16x4) && __is_same(ET, f16)) {
        return post_op4_lds_f16(v, nl, prm, bn, has_bias, has_scale, slope);
    } else {
        typedef Elem<ET> E;
        if (has_bias) v += *reinterpret_cast<const f32x4*>(prm + nl);
        typename E::v4 h = E::down(v);
        if (has_scale) h = E::down(E::up(h) * *reinterpret_cast<const f32x4*>(prm + bn + nl) + *reinterpret_cast<const f32x4*>(prm + 2 * bn + nl));
        const f32x4 hf = E::up(h);
        const f32x4 tf = E::up(E::down(hf * slope));
        return E::down(__builtin_elementwise_max(hf, tf));  // both operands are element-type values: the maximum is one of them
    }
}

// Post-ops on four consecutive channels n..n+3 of one pixel (n % 4 == 0, n + 3 < Cout).
__device__ __forceinline__ f16x4 post_op4(f32x4 v, int n, const ConvArgs& a) {
    if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + n);
    f16x4 h = __builtin_convertvector(v, f16x4);
    if (a.scale) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + n), sh = *reinterpret_cast<const f32x4*>(a.shift + n);
        h = __builtin_convertvector(__builtin_convertvector(h, f32x4) * sc + sh, f16x4);
    }
    return leaky4_f16(h, a.slope);
}


// v_mfma_f32_16x16x32: D (16 x 16, lane l holds column l & 15, rows 4 * (l >> 4) .. + 3) += A (16 x 32, lane l: row l & 15,
// k = 8 * (l >> 4) .. + 7) * B (32 x 16, lane l: column l & 15, the same k).
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

