// Shared declarations of the liby5hip.so translation units (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/y5hip.h"

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// The two 16-bit element types of the activation / weight tensors (y5_conv_desc.dtype: Y5_F16, Y5_BF16).  Both accumulate
// in fp32; every tensor operation rounds its result to the element type once, like torch's half / bfloat16 kernels.
// Mixed-precision helpers on a pair of fp16 values packed in 32 bits: fp16 inputs widened, fp32 arithmetic, result rounded
// to fp16 - the conversion / multiply / conversion sequence of the layer epilogues in one instruction per value.
static __device__ __forceinline__ unsigned mix_mul2(unsigned h, float s, float negzero) {  // fp16(float(h) * s); negzero = -0.f keeps the sign of a zero product
    unsigned t;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(t)
        : "v"(h), "v"(s), "v"(negzero));
    return t;
}
// max of two fp16 pairs (the compiler's fmax first re-rounds both operands to quiet signalling NaNs: three instructions)
static __device__ __forceinline__ unsigned pk_max2(unsigned a, unsigned b) {
    unsigned t;
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(t) : "v"(a), "v"(b));
    return t;
}
// LeakyReLU on four fp16 values as max(h, fp16(float(h) * slope)), slope in (0, 1] (1 = no activation)
static __device__ __forceinline__ f16x4 leaky4_f16(f16x4 h, float slope) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 u = __builtin_bit_cast(u32x2, h);
    const float nz = -0.f;
    const u32x2 r = {pk_max2(u[0], mix_mul2(u[0], slope, nz)), pk_max2(u[1], mix_mul2(u[1], slope, nz))};
    return __builtin_bit_cast(f16x4, r);
}

template <typename T>
struct Elem;
template <>
struct Elem<f16> {
    typedef f16x4 v4;
    typedef f16x8 v8;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ v4 down(f32x4 v) { return __builtin_convertvector(v, f16x4); }
    static __device__ __forceinline__ f32x4 up(v4 h) { return __builtin_convertvector(h, f32x4); }
    static __device__ __forceinline__ v8 add(v8 a, v8 b) { return a + b; }
    static __device__ __forceinline__ v8 max8(v8 a, v8 b) { return __builtin_elementwise_max(a, b); }
};
template <>
struct Elem<bf16> {
    typedef bf16x4 v4;
    typedef bf16x8 v8;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ v4 down(f32x4 v) { return __builtin_convertvector(v, bf16x4); }
    static __device__ __forceinline__ f32x4 up(v4 h) { return __builtin_convertvector(h, f32x4); }
    static __device__ __forceinline__ v8 add(v8 a, v8 b) {
        return __builtin_convertvector(__builtin_convertvector(a, f32x8) + __builtin_convertvector(b, f32x8), bf16x8);
    }
    static __device__ __forceinline__ v8 max8(v8 a, v8 b) {
        return __builtin_convertvector(__builtin_elementwise_max(__builtin_convertvector(a, f32x8), __builtin_convertvector(b, f32x8)), bf16x8);
    }
};

// Records a failure for y5_last_error() and returns `code`.
int y5_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define Y5_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return y5_fail(Y5_EHIP, "%s -> %s", #expr, hipGetErrorString(e_));  \
    } while (0)

#define Y5_REQUIRE(cond, ...)                              \
    do {                                                   \
        if (!(cond)) return y5_fail(Y5_EINVAL, __VA_ARGS__); \
    } while (0)

// Checks the launch that was just issued.
#define Y5_LAUNCH_CHECK(name)                                                                      \
    do {                                                                                           \
        hipError_t e_ = hipGetLastError();                                                         \
        if (e_ != hipSuccess) return y5_fail(Y5_EHIP, "launch %s -> %s", name, hipGetErrorString(e_)); \
    } while (0)

// Ordinal of the current device, folded into 0..63 (index of the per-device "done once" bits).
static inline int y5_current_device() {
    int d = 0;
    (void)hipGetDevice(&d);
    return d & 63;
}

static inline bool y5_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Internal launchers (one per kernel family); the extern "C" entry points and the plan replay call these.
// Optional per-channel sums of the stored outputs, accumulated by the convolution's epilogue (im2col kernel family only):
// partials[r][off + n] += sum, partials[r][c + off + n] += sum of squares over the pixels of M-tile r modulo rows.
struct Y5ConvStats {
    float* partials;
    int rows, c, off;
};
int y5_launch_conv(const y5_conv_desc& d, hipStream_t s, const Y5ConvStats* st = nullptr);
// 3x3 stride-1 halo-tile kernel (conv_halo.hip): returns 1 and sets *rc when it took the convolution.
int y5_try_conv_halo(const y5_conv_desc& d, hipStream_t s, int force_bn, int* rc);
int y5_try_conv_ws1x1(const y5_conv_desc& d, hipStream_t s, int which, int* rc);
int y5_try_conv3x3_ws(const y5_conv_desc& d, hipStream_t s, int which, int* rc);
int y5_try_conv_lw(const y5_conv_desc& d, hipStream_t s, int which, int* rc);
int y5_conv_lw_count();
// 3x3 stride-1 kernel with the pixel operand resident in LDS (conv_lin3.hip)
int y5_try_conv_lin(const y5_conv_desc& d, hipStream_t s, int which, int* rc);
int y5_conv_lin_count();
// 3x3 stride-2 32 -> 64 persistent kernel (conv_s2c32.hip)
int y5_try_conv_s2c32(const y5_conv_desc& d, hipStream_t s, int which, int* rc);
int y5_conv_s2c32_count();
// 3x3 stride-2 64 -> 128 persistent kernel (conv_s2c64.hip): one variant, numbered before the two above
int y5_try_conv_s2c64(const y5_conv_desc& d, hipStream_t s, int* rc);
int y5_conv_s2c64_count();
// 1x1 persistent kernel for 128 / 256 channels (conv_p1x1.hip): two variants, numbered before the stride-2 ones
int y5_try_conv_p1x1(const y5_conv_desc& d, hipStream_t s, int which, int* rc);
int y5_conv_p1x1_count();
int y5_try_conv_p1x1_chain(const y5_conv_desc& f, const y5_conv_desc& g, int keep_first, hipStream_t s, int* rc);
int y5_try_conv_p1x1_cat2(const y5_conv_desc& d, const void* x2, int x2_ld, int cin2, hipStream_t s, int* rc);
int y5_launch_detect(const y5_conv_desc& c, const y5_decode_desc& d, int variant, hipStream_t s);
int y5_detect_variant_count();
int y5_try_detect_persistent(const y5_conv_desc& c, const y5_decode_desc& d, hipStream_t s, int* rc);
int y5_launch_conv_chain(const y5_conv_desc& f, const y5_conv_desc& g, int keep_first, int variant, hipStream_t s);
int y5_try_conv_c64_chain(const y5_conv_desc& f, const y5_conv_desc& g, int keep_first, hipStream_t s, int* rc);
int y5_try_conv_c128_chain(const y5_conv_desc& f, const y5_conv_desc& g, int keep_first, hipStream_t s, int* rc);
int y5_conv_variant_count();
// phased implicit-GEMM kernels for the compute-bound layers (conv_ph.hip): the last fp16 variants, and the bf16 variants
// behind the im2col tile configurations
int y5_try_conv_ph(const y5_conv_desc& d, hipStream_t s, int which, int* rc);
int y5_conv_ph_count();
// BottleneckCSP with one Bottleneck as one kernel (conv_csp.hip)
int y5_launch_csp(const y5_csp_desc& d, hipStream_t s);
int y5_conv_bf16_variant_count();
int y5_launch_conv_cat2(const y5_conv_desc& d, const void* x2, int x2_ld, int cin2, hipStream_t s);
int y5_conv_cat2_variant_count();
// Focus gather + 3x3 convolution from the NCHW image (conv_focus.hip); d.x is the image, d.cin its 3 channels.
// input: Y5_IN_F16 / Y5_IN_U8 (NCHW planes) or Y5_IN_U8_HWC / Y5_IN_U8_HWC_BGR (interleaved bytes)
int y5_launch_focus_conv(const y5_conv_desc& d, hipStream_t s, int input = Y5_IN_F16);
int y5_launch_stem_conv(const y5_conv_desc& d, hipStream_t s, bool u8 = false);
// Focus + conv and the following 3x3 stride-2 conv as one kernel (conv_stem2.hip)
int y5_launch_focus_conv2(const y5_conv_desc& f, const y5_conv_desc& g, hipStream_t s);
int y5_launch_normalize_u8(const void* x, void* y, long long n, hipStream_t s);
int y5_launch_pack_image(const void* x, void* y, int batch, int c, int h, int w, int focus, int y_ld, hipStream_t s);
int y5_launch_maxpool(const void* x, int x_ld, void* y, int y_ld, int batch, int h, int w, int c, int k, hipStream_t s, int dtype = Y5_F16);
int y5_launch_maxpool_cascade(const void* x, int x_ld, void* y, int y_ld, int y_step, int levels, int batch, int h, int w, int c,
                              int k, hipStream_t s, int dtype = Y5_F16);
int y5_launch_upsample2x(const void* x, int x_ld, void* y, int y_ld, int batch, int h, int w, int c, int mode,
                         hipStream_t s, int dtype = Y5_F16);
int y5_launch_copy_channels(const void* x, int x_ld, void* y, int y_ld, int64_t pixels, int c, hipStream_t s);
int y5_launch_decode(const y5_decode_desc& d, hipStream_t s);
int y5_launch_decode_multi(const y5_decode_desc* d, int n, hipStream_t s);
int y5_launch_sigmoid(const void* x, void* y, long long n, int libm, hipStream_t s);
int y5_bn_stats_row_count(long long pixels, int c);
int y5_launch_bn_stats(const void* x, int x_ld, long long pixels, int c, float* partials, hipStream_t s);
int y5_launch_bn_finalize(const float* partials, int rows, long long count, int c, int c_valid, const float* gamma, const float* beta,
                          float eps, float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                          hipStream_t s);
int y5_launch_affine_act(const void* x, int x_ld, void* y, int y_ld, const void* res, int res_ld, long long pixels, int c,
                         const float* scale, const float* shift, int act, float slope, hipStream_t s);
