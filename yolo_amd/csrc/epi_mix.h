// Epilogue pieces of the persistent tile kernels (conv_csp.hip, conv_c64.hip): the accumulator-to-channel order of the
// MFMA tiles and the per-layer post-ops in registers.  Included inside an anonymous namespace after conv_common.h.
#pragma once

// MFMA row i of a 32-row weight fragment holds output channel perm(i): lane (lr, lh) then owns channels [16*lh, 16*lh + 16)
// of pixel lr in accumulator order, i.e. two whole 16-byte pieces of a [pixel][channel] row.
__device__ __forceinline__ int perm_row(int i) { return 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3); }

// (mix_mul2 and pk_max2 come from conv_common.h.)  fp16(float(h) * s + b) on a pair of fp16 values, one rounding each:
__device__ __forceinline__ unsigned mix_fma2(unsigned h, float s0, float s1, float b0, float b1) {  // fp16(float(h) * s + b)
    unsigned t;
    asm("v_fma_mixlo_f16 %0, %1, %2, %4 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, %3, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(t)
        : "v"(h), "v"(s0), "v"(s1), "v"(b0), "v"(b1));
    return t;
}

// Post-op parameters of the 16 channels nl..nl+15 a lane owns, of a stage with n channels (LDS -> registers).
template <int F>
struct Prm16 {
    f32x4 b[(F & 1) ? 4 : 1], sc[(F & 2) ? 4 : 1], sh[(F & 2) ? 4 : 1];
    __device__ __forceinline__ void load(const float* prm, int n, int nl) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if constexpr (F & 1) b[g] = *reinterpret_cast<const f32x4*>(prm + nl + 4 * g);
            if constexpr (F & 2) {
                sc[g] = *reinterpret_cast<const f32x4*>(prm + n + nl + 4 * g);
                sh[g] = *reinterpret_cast<const f32x4*>(prm + 2 * n + nl + 4 * g);
            }
        }
    }
};

// The 16 channels a lane owns of one accumulator tile after the layer's post-ops, rounded exactly as y5_conv_f16 rounds
// them: fp16(acc + bias), fp16(that * scale + shift), LeakyReLU as max(h, fp16(h * slope)).
template <int F>
__device__ __forceinline__ void finish(const f32x16& acc, const Prm16<F>& q, float slope, float negzero, f16x8& lo, f16x8& hi) {
    unsigned o[8];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        if constexpr (F & 1) v += q.b[g];
        const f16x4 c = __builtin_convertvector(v, f16x4);
        unsigned c0 = __builtin_bit_cast(unsigned, __builtin_shufflevector(c, c, 0, 1));
        unsigned c1 = __builtin_bit_cast(unsigned, __builtin_shufflevector(c, c, 2, 3));
        if constexpr (F & 2) {
            c0 = mix_fma2(c0, q.sc[g][0], q.sc[g][1], q.sh[g][0], q.sh[g][1]);
            c1 = mix_fma2(c1, q.sc[g][2], q.sc[g][3], q.sh[g][2], q.sh[g][3]);
        }
        const unsigned t0 = mix_mul2(c0, slope, negzero), t1 = mix_mul2(c1, slope, negzero);
        o[2 * g] = pk_max2(c0, t0);
        o[2 * g + 1] = pk_max2(c1, t1);
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    lo = __builtin_bit_cast(f16x8, (u32x4){o[0], o[1], o[2], o[3]});
    hi = __builtin_bit_cast(f16x8, (u32x4){o[4], o[5], o[6], o[7]});
}
template <int F>
__device__ __forceinline__ void finish(const f32x16& acc, const float* prm, int n, int nl, float slope, float negzero, f16x8& lo, f16x8& hi) {
    Prm16<F> q;
    q.load(prm, n, nl);
    finish<F>(acc, q, slope, negzero, lo, hi);
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int e = 0; e < 16; ++e) z[e] = 0.f;
    return z;
}

