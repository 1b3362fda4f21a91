// extern "C" surface of liby5hip.so: error reporting, single-op entry points and the launch plan.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "y5_common.h"

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

int y5_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* y5_last_error(void) { return g_err; }
extern "C" int y5_version(void) { return 100; }

extern "C" int y5_device_arch(char* buf, int n) {
    Y5_REQUIRE(buf && n > 0, "device_arch: bad buffer");
    int dev = 0;
    Y5_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    Y5_HIP(hipGetDeviceProperties(&prop, dev));
    snprintf(buf, (size_t)n, "%s", prop.gcnArchName);
    return Y5_OK;
}

// ------------------------------------------------------------------------------------------
// single-op entry points
// ------------------------------------------------------------------------------------------
extern "C" int y5_conv_f16(const y5_conv_desc* d, y5_stream_t stream) {
    Y5_REQUIRE(d, "conv: null descriptor");
    return y5_launch_conv(*d, static_cast<hipStream_t>(stream));
}
extern "C" int y5_focus_conv_f16(const y5_conv_desc* d, y5_stream_t stream) {
    Y5_REQUIRE(d, "focus_conv: null descriptor");
    return y5_launch_focus_conv(*d, static_cast<hipStream_t>(stream));
}

extern "C" int y5_sigmoid_f16(const void* x, void* y, int64_t n, int libm, y5_stream_t stream) {
    return y5_launch_sigmoid(x, y, n, libm, static_cast<hipStream_t>(stream));
}
extern "C" int y5_stem_conv_f16(const y5_conv_desc* d, y5_stream_t stream) {
    Y5_REQUIRE(d, "stem_conv: null descriptor");
    return y5_launch_stem_conv(*d, static_cast<hipStream_t>(stream));
}

extern "C" int y5_stem_conv_u8(const y5_conv_desc* d, y5_stream_t stream) {
    Y5_REQUIRE(d, "stem_conv_u8: null descriptor");
    return y5_launch_stem_conv(*d, static_cast<hipStream_t>(stream), true);
}

extern "C" int y5_focus_conv_u8(const y5_conv_desc* d, y5_stream_t stream) {
    Y5_REQUIRE(d, "focus_conv_u8: null descriptor");
    return y5_launch_focus_conv(*d, static_cast<hipStream_t>(stream), true);
}

extern "C" int y5_focus_conv2_f16(const y5_conv_desc* focus, const y5_conv_desc* conv, y5_stream_t stream) {
    Y5_REQUIRE(focus && conv, "focus_conv2: null descriptor");
    return y5_launch_focus_conv2(*focus, *conv, static_cast<hipStream_t>(stream));
}

extern "C" int y5_focus_conv_u8_hwc(const y5_conv_desc* d, int input, y5_stream_t stream) {
    Y5_REQUIRE(d, "focus_conv: null descriptor");
    Y5_REQUIRE(input == Y5_IN_U8_HWC || input == Y5_IN_U8_HWC_BGR, "focus_conv_u8_hwc: input must be Y5_IN_U8_HWC or Y5_IN_U8_HWC_BGR (got %d)", input);
    return y5_launch_focus_conv(*d, static_cast<hipStream_t>(stream), input);
}
extern "C" int y5_normalize_u8_f16(const void* x, void* y, int64_t n, y5_stream_t stream) {
    return y5_launch_normalize_u8(x, y, n, static_cast<hipStream_t>(stream));
}
extern "C" int y5_conv_num_variants(void) { return y5_conv_variant_count(); }
extern "C" int y5_conv_num_variants_bf16(void) { return y5_conv_bf16_variant_count(); }
extern "C" int y5_conv_num_variants_phased(void) { return y5_conv_ph_count(); }
extern "C" int y5_pack_image_f16(const void* x, void* y, int batch, int c, int h, int w, int focus, int y_ld,
                                 y5_stream_t stream) {
    return y5_launch_pack_image(x, y, batch, c, h, w, focus, y_ld, static_cast<hipStream_t>(stream));
}
extern "C" int y5_maxpool_f16(const void* x, int x_ld, void* y, int y_ld, int batch, int h, int w, int c, int k,
                              y5_stream_t stream) {
    return y5_launch_maxpool(x, x_ld, y, y_ld, batch, h, w, c, k, static_cast<hipStream_t>(stream));
}
extern "C" int y5_maxpool_cascade_f16(const void* x, int x_ld, void* y, int y_ld, int y_step, int levels, int batch, int h, int w,
                                      int c, int k, y5_stream_t stream) {
    return y5_launch_maxpool_cascade(x, x_ld, y, y_ld, y_step, levels, batch, h, w, c, k, static_cast<hipStream_t>(stream));
}
extern "C" int y5_maxpool_x16(const void* x, int x_ld, void* y, int y_ld, int batch, int h, int w, int c, int k, int dtype, y5_stream_t stream) {
    return y5_launch_maxpool(x, x_ld, y, y_ld, batch, h, w, c, k, static_cast<hipStream_t>(stream), dtype);
}
extern "C" int y5_maxpool_cascade_x16(const void* x, int x_ld, void* y, int y_ld, int y_step, int levels, int batch, int h, int w, int c,
                                      int k, int dtype, y5_stream_t stream) {
    return y5_launch_maxpool_cascade(x, x_ld, y, y_ld, y_step, levels, batch, h, w, c, k, static_cast<hipStream_t>(stream), dtype);
}
extern "C" int y5_upsample2x_x16(const void* x, int x_ld, void* y, int y_ld, int batch, int h, int w, int c, int mode, int dtype,
                                 y5_stream_t stream) {
    return y5_launch_upsample2x(x, x_ld, y, y_ld, batch, h, w, c, mode, static_cast<hipStream_t>(stream), dtype);
}
extern "C" int y5_upsample2x_f16(const void* x, int x_ld, void* y, int y_ld, int batch, int h, int w, int c, int mode,
                                 y5_stream_t stream) {
    return y5_launch_upsample2x(x, x_ld, y, y_ld, batch, h, w, c, mode, static_cast<hipStream_t>(stream));
}
extern "C" int y5_conv_stats_f16(const y5_conv_desc* d, float* partials, int rows, int stats_c, int stats_off, y5_stream_t stream) {
    Y5_REQUIRE(d, "conv_stats: null descriptor");
    const Y5ConvStats st{partials, rows, stats_c, stats_off};
    return y5_launch_conv(*d, static_cast<hipStream_t>(stream), &st);
}

extern "C" int y5_bn_stats_rows(int64_t pixels, int c) { return y5_bn_stats_row_count(pixels, c); }

extern "C" int y5_bn_stats_f16(const void* x, int x_ld, int64_t pixels, int c, float* partials, y5_stream_t stream) {
    return y5_launch_bn_stats(x, x_ld, pixels, c, partials, static_cast<hipStream_t>(stream));
}

extern "C" int y5_bn_finalize_f32(const float* partials, int rows, int64_t count, int c, int c_valid, const float* gamma,
                                  const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* scale,
                                  float* shift, y5_stream_t stream) {
    return y5_launch_bn_finalize(partials, rows, count, c, c_valid, gamma, beta, eps, momentum, running_mean, running_var, scale, shift,
                                 static_cast<hipStream_t>(stream));
}

extern "C" int y5_affine_act_f16(const void* x, int x_ld, void* y, int y_ld, const void* res, int res_ld, int64_t pixels, int c,
                                 const float* scale, const float* shift, int act, float slope, y5_stream_t stream) {
    return y5_launch_affine_act(x, x_ld, y, y_ld, res, res_ld, pixels, c, scale, shift, act, slope, static_cast<hipStream_t>(stream));
}

extern "C" int y5_copy_channels_f16(const void* x, int x_ld, void* y, int y_ld, int64_t pixels, int c,
                                    y5_stream_t stream) {
    return y5_launch_copy_channels(x, x_ld, y, y_ld, pixels, c, static_cast<hipStream_t>(stream));
}
extern "C" int y5_detect_decode_f16(const y5_decode_desc* d, y5_stream_t stream) {
    Y5_REQUIRE(d, "decode: null descriptor");
    return y5_launch_decode(*d, static_cast<hipStream_t>(stream));
}

extern "C" int y5_detect_decode_multi_f16(const y5_decode_desc* levels, int n_levels, y5_stream_t stream) {
    return y5_launch_decode_multi(levels, n_levels, static_cast<hipStream_t>(stream));
}

extern "C" int y5_detect_f16(const y5_conv_desc* conv, const y5_decode_desc* d, int variant, y5_stream_t stream) {
    Y5_REQUIRE(conv && d, "detect: null descriptor");
    return y5_launch_detect(*conv, *d, variant, static_cast<hipStream_t>(stream));
}

extern "C" int y5_detect_num_variants(void) { return y5_detect_variant_count(); }

extern "C" int y5_conv_chain_f16(const y5_conv_desc* first, const y5_conv_desc* second, int keep_first, int variant, y5_stream_t stream) {
    Y5_REQUIRE(first && second, "conv chain: null descriptor");
    return y5_launch_conv_chain(*first, *second, keep_first, variant, static_cast<hipStream_t>(stream));
}

extern "C" int y5_conv_chain_num_variants(void) { return y5_conv_lw_count() + 3; }

extern "C" int y5_conv_cat2_f16(const y5_conv_desc* d, const void* x2, int x2_ld, int cin2, y5_stream_t stream) {
    Y5_REQUIRE(d, "conv_cat2: null descriptor");
    return y5_launch_conv_cat2(*d, x2, x2_ld, cin2, static_cast<hipStream_t>(stream));
}

extern "C" int y5_conv_cat2_num_variants(void) { return y5_conv_cat2_variant_count(); }

extern "C" int y5_csp_f16(const y5_csp_desc* d, y5_stream_t stream) {
    Y5_REQUIRE(d, "csp: null descriptor");
    return y5_launch_csp(*d, static_cast<hipStream_t>(stream));
}

// ------------------------------------------------------------------------------------------
// launch plan
// ------------------------------------------------------------------------------------------
namespace {

struct SlotRange {
    const unsigned char* base = nullptr;
    size_t bytes = 0;
};

struct PtrFix {  // pointer field of an op that must be rebased on an external slot at run time
    int field;   // index into the op's pointer table
    int slot;
    size_t offset;
};

struct Op {
    int kind = 0;
    y5_conv_desc conv{};
    y5_conv_desc conv2{};  // Y5_OP_CONV_CHAIN: the second convolution; i[0] variant, i[1] keep_first
                           // Y5_OP_CONV_CAT2: x = the second source, x_ld its pitch, cin its channels
    y5_csp_desc csp{};     // Y5_OP_CSP
    y5_decode_desc dec{};
    y5_decode_desc decm[Y5_MAX_LEVELS] = {};  // Y5_OP_DECODE_MULTI: i[0] levels
    // generic pointwise arguments
    const void* px = nullptr;
    void* py = nullptr;
    int i[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long pixels = 0;
    std::vector<PtrFix> fixes;
    double flops = 0.0, bytes = 0.0;
    int flags = 0;  // Y5_PLAN_SIDE / Y5_PLAN_JOIN
    int dtype = Y5_F16;  // pointwise ops: element type
};

}  // namespace

struct y5_plan {
    std::vector<Op> ops;
    int dtype = Y5_F16;  // element type of the pointwise ops recorded from now on (y5_plan_set_dtype)
    SlotRange slots[16];
    // side stream for ops marked Y5_PLAN_SIDE and the events that order it against the caller's stream; created on
    // first use on the device that is current then
    hipStream_t side = nullptr;
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;
    ~y5_plan() {
        if (fork_ev) (void)hipEventDestroy(fork_ev);
        if (join_ev) (void)hipEventDestroy(join_ev);
        if (side) (void)hipStreamDestroy(side);
    }
};

namespace {

// If p lies inside a bound slot, remember to rebase it; returns false for a pointer straddling a slot end.
void note_ptr(y5_plan* plan, Op& op, int field, const void* p) {
    if (!p) return;
    const unsigned char* c = static_cast<const unsigned char*>(p);
    for (int s = 0; s < 16; ++s) {
        const SlotRange& r = plan->slots[s];
        if (r.base && c >= r.base && c < r.base + r.bytes) {
            op.fixes.push_back({field, s, (size_t)(c - r.base)});
            return;
        }
    }
}

bool set_ptr(Op& op, int field, const void* v) {
    void* nv = const_cast<void*>(v);
    switch (op.kind) {
        case Y5_OP_CONV:
        case Y5_OP_FOCUS_CONV:
        case Y5_OP_FOCUS_CONV_U8:
        case Y5_OP_FOCUS_CONV_HWC:
        case Y5_OP_STEM_CONV:
        case Y5_OP_STEM_CONV_U8:
            if (field == 0) op.conv.x = v;
            else if (field == 1) op.conv.y = nv;
            else if (field == 2) op.conv.res = v;
            else return false;
            return true;
        case Y5_OP_DECODE_MULTI: {
            const int l = field / 3, f = field % 3;
            if (l >= op.i[0]) return false;
            if (f == 0) op.decm[l].head = v;
            else if (f == 1) op.decm[l].x_out = nv;
            else op.decm[l].z_out = nv;
            return true;
        }
        case Y5_OP_FOCUS_CONV2:
            if (field == 0) op.conv.x = v;
            else if (field == 1) op.conv2.y = nv;
            else return false;
            return true;
        case Y5_OP_CONV_CHAIN:
            if (field == 0) op.conv.x = v;
            else if (field == 1) op.conv.y = nv;
            else if (field == 2) op.conv.res = v;
            else if (field == 3) op.conv2.y = nv;
            else return false;
            return true;
        case Y5_OP_CONV_CAT2:
            if (field == 0) op.conv.x = v;
            else if (field == 1) op.conv.y = nv;
            else if (field == 2) op.conv.res = v;
            else if (field == 3) op.conv2.x = v;
            else return false;
            return true;
        case Y5_OP_CSP:
            if (field == 0) op.csp.x = v;
            else if (field == 1) op.csp.y = nv;
            else return false;
            return true;
        case Y5_OP_DETECT:
            if (field == 0) op.conv.x = v;
            else if (field == 1) op.dec.x_out = nv;
            else if (field == 2) op.dec.z_out = nv;
            else return false;
            return true;
        case Y5_OP_DECODE:
            if (field == 0) op.dec.head = v;
            else if (field == 1) op.dec.x_out = nv;
            else if (field == 2) op.dec.z_out = nv;
            else return false;
            return true;
        default:
            if (field == 0) op.px = v;
            else if (field == 1) op.py = nv;
            else return false;
            return true;
    }
}

int run_op(const Op& op, hipStream_t s) {
    switch (op.kind) {
        case Y5_OP_CONV: return y5_launch_conv(op.conv, s);
        case Y5_OP_FOCUS_CONV: return y5_launch_focus_conv(op.conv, s);
        case Y5_OP_FOCUS_CONV_U8: return y5_launch_focus_conv(op.conv, s, Y5_IN_U8);
        case Y5_OP_FOCUS_CONV_HWC: return y5_launch_focus_conv(op.conv, s, op.i[0]);
        case Y5_OP_STEM_CONV: return y5_launch_stem_conv(op.conv, s);
        case Y5_OP_STEM_CONV_U8: return y5_launch_stem_conv(op.conv, s, true);
        case Y5_OP_NORMALIZE_U8: return y5_launch_normalize_u8(op.px, op.py, op.pixels, s);
        case Y5_OP_PACK: return y5_launch_pack_image(op.px, op.py, op.i[0], op.i[1], op.i[2], op.i[3], op.i[4], op.i[5], s);
        case Y5_OP_MAXPOOL: return y5_launch_maxpool(op.px, op.i[0], op.py, op.i[1], op.i[2], op.i[3], op.i[4], op.i[5], op.i[6], s, op.dtype);
        case Y5_OP_MAXPOOL_CASCADE:
            return y5_launch_maxpool_cascade(op.px, op.i[0], op.py, op.i[1], op.i[2], op.i[3], op.i[4], op.i[5], op.i[6], op.i[7], (int)op.pixels, s, op.dtype);
        case Y5_OP_UPSAMPLE: return y5_launch_upsample2x(op.px, op.i[0], op.py, op.i[1], op.i[2], op.i[3], op.i[4], op.i[5], op.i[6], s, op.dtype);
        case Y5_OP_COPY: return y5_launch_copy_channels(op.px, op.i[0], op.py, op.i[1], op.pixels, op.i[2], s);
        case Y5_OP_DECODE: return y5_launch_decode(op.dec, s);
        case Y5_OP_DECODE_MULTI: return y5_launch_decode_multi(op.decm, op.i[0], s);
        case Y5_OP_DETECT: return y5_launch_detect(op.conv, op.dec, op.i[0], s);
        case Y5_OP_CONV_CHAIN: return y5_launch_conv_chain(op.conv, op.conv2, op.i[1], op.i[0], s);
        case Y5_OP_CONV_CAT2: return y5_launch_conv_cat2(op.conv, op.conv2.x, op.conv2.x_ld, op.conv2.cin, s);
        case Y5_OP_CSP: return y5_launch_csp(op.csp, s);
        case Y5_OP_FOCUS_CONV2: return y5_launch_focus_conv2(op.conv, op.conv2, s);
    }
    return y5_fail(Y5_EINVAL, "plan: unknown op kind %d", op.kind);
}

// Returns the op with external pointers substituted.
int resolve(const y5_plan* p, size_t idx, void* const* ext, int n_ext, Op& out) {
    out = p->ops[idx];
    for (const PtrFix& f : out.fixes) {
        if (f.slot >= n_ext || !ext || !ext[f.slot])
            return y5_fail(Y5_EINVAL, "plan: op %zu needs external slot %d but only %d were passed", idx, f.slot, n_ext);
        if (!set_ptr(out, f.field, static_cast<const unsigned char*>(ext[f.slot]) + f.offset))
            return y5_fail(Y5_EINVAL, "plan: op %zu has no pointer field %d", idx, f.field);
    }
    return Y5_OK;
}

template <typename F>
int guarded(F&& f) {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        return y5_fail(Y5_ENOMEM, "plan: host allocation failed");
    } catch (...) {
        return y5_fail(Y5_EINVAL, "plan: unexpected C++ exception");
    }
}

}  // namespace

extern "C" int y5_plan_create(y5_plan** out) {
    Y5_REQUIRE(out, "plan_create: null output");
    y5_plan* p = new (std::nothrow) y5_plan();
    if (!p) return y5_fail(Y5_ENOMEM, "plan_create: host allocation failed");
    *out = p;
    return Y5_OK;
}

extern "C" void y5_plan_destroy(y5_plan* p) { delete p; }

extern "C" int y5_plan_bind_slot(y5_plan* p, int slot, const void* base, size_t bytes) {
    Y5_REQUIRE(p, "plan_bind_slot: null plan");
    Y5_REQUIRE(slot >= 0 && slot < 16, "plan_bind_slot: slot %d out of range 0..15", slot);
    Y5_REQUIRE(base && bytes > 0, "plan_bind_slot: empty range");
    p->slots[slot].base = static_cast<const unsigned char*>(base);
    p->slots[slot].bytes = bytes;
    return Y5_OK;
}

extern "C" int y5_plan_set_dtype(y5_plan* p, int dtype) {
    Y5_REQUIRE(p, "plan_set_dtype: null plan");
    Y5_REQUIRE(dtype == Y5_F16 || dtype == Y5_BF16, "plan_set_dtype: dtype %d not in {Y5_F16, Y5_BF16}", dtype);
    p->dtype = dtype;
    return Y5_OK;
}

extern "C" int y5_plan_add_conv(y5_plan* p, const y5_conv_desc* d) {
    Y5_REQUIRE(p && d, "plan_add_conv: null argument");
    return guarded([&]() {
        Op op;
        op.kind = Y5_OP_CONV;
        op.conv = *d;
        note_ptr(p, op, 0, d->x);
        note_ptr(p, op, 1, d->y);
        note_ptr(p, op, 2, d->res);
        const double m = (double)d->batch * d->ho * d->wo;
        op.flops = 2.0 * m * d->cout * d->ksize * d->ksize * d->cin;
        op.bytes = 2.0 * ((double)d->batch * d->h * d->w_in * d->cin + m * d->cout * (d->res ? 2 : 1) +
                          (double)d->cout * d->ksize * d->ksize * d->cin);
        p->ops.push_back(op);
        return Y5_OK;
    });
}

static int add_focus_conv(y5_plan* p, const y5_conv_desc* d, int kind);
extern "C" int y5_plan_add_focus_conv(y5_plan* p, const y5_conv_desc* d) { return add_focus_conv(p, d, Y5_OP_FOCUS_CONV); }
extern "C" int y5_plan_add_focus_conv_u8(y5_plan* p, const y5_conv_desc* d) { return add_focus_conv(p, d, Y5_OP_FOCUS_CONV_U8); }
extern "C" int y5_plan_add_focus_conv_u8_hwc(y5_plan* p, const y5_conv_desc* d, int input) {
    Y5_REQUIRE(input == Y5_IN_U8_HWC || input == Y5_IN_U8_HWC_BGR, "plan_add_focus_conv_u8_hwc: input must be Y5_IN_U8_HWC or Y5_IN_U8_HWC_BGR (got %d)", input);
    const int rc = add_focus_conv(p, d, Y5_OP_FOCUS_CONV_HWC);
    if (rc == Y5_OK) p->ops.back().i[0] = input;
    return rc;
}
extern "C" int y5_plan_add_stem_conv(y5_plan* p, const y5_conv_desc* d) { return add_focus_conv(p, d, Y5_OP_STEM_CONV); }
extern "C" int y5_plan_add_stem_conv_u8(y5_plan* p, const y5_conv_desc* d) { return add_focus_conv(p, d, Y5_OP_STEM_CONV_U8); }

static int add_focus_conv(y5_plan* p, const y5_conv_desc* d, int kind) {
    Y5_REQUIRE(p && d, "plan_add_focus_conv: null argument");
    return guarded([&]() {
        Op op;
        op.kind = kind;
        op.conv = *d;
        note_ptr(p, op, 0, d->x);
        note_ptr(p, op, 1, d->y);
        const double m = (double)d->batch * d->ho * d->wo;
        const bool stem = kind == Y5_OP_STEM_CONV || kind == Y5_OP_STEM_CONV_U8;
        const bool u8 = kind == Y5_OP_FOCUS_CONV_U8 || kind == Y5_OP_STEM_CONV_U8 || kind == Y5_OP_FOCUS_CONV_HWC;
        const double taps = stem ? 9.0 : 9.0 * 4;  // Focus: 3x3 over the four space-to-depth copies of each channel
        op.flops = 2.0 * m * d->cout * taps * d->cin;
        op.bytes = (u8 ? 1.0 : 2.0) * d->batch * d->h * d->w_in * d->cin + 2.0 * (m * d->cout + (double)d->cout * taps * d->cin);
        p->ops.push_back(op);
        return Y5_OK;
    });
}

static int add_pointwise(y5_plan* p, int kind, const void* x, void* y, const int* iv, int ni, long long pixels, double bytes) {
    return guarded([&]() {
        Op op;
        op.kind = kind;
        op.px = x;
        op.py = y;
        for (int k = 0; k < ni; ++k) op.i[k] = iv[k];
        op.pixels = pixels;
        op.bytes = bytes;
        op.dtype = p->dtype;
        note_ptr(p, op, 0, x);
        note_ptr(p, op, 1, y);
        p->ops.push_back(op);
        return Y5_OK;
    });
}

extern "C" int y5_plan_add_pack_image(y5_plan* p, const void* x, void* y, int batch, int c, int h, int w, int focus, int y_ld) {
    Y5_REQUIRE(p, "plan_add_pack_image: null plan");
    const int iv[6] = {batch, c, h, w, focus, y_ld};
    const double px = (double)batch * h * w;
    return add_pointwise(p, Y5_OP_PACK, x, y, iv, 6, 0, 2.0 * px * c + 2.0 * (focus ? px / 4 : px) * y_ld);
}

extern "C" int y5_plan_add_normalize_u8(y5_plan* p, const void* x, void* y, int64_t n) {
    Y5_REQUIRE(p, "plan_add_normalize_u8: null plan");
    return add_pointwise(p, Y5_OP_NORMALIZE_U8, x, y, nullptr, 0, n, 3.0 * (double)n);
}

extern "C" int y5_plan_add_maxpool(y5_plan* p, const void* x, int x_ld, void* y, int y_ld, int batch, int h, int w, int c, int k) {
    Y5_REQUIRE(p, "plan_add_maxpool: null plan");
    const int iv[7] = {x_ld, y_ld, batch, h, w, c, k};
    return add_pointwise(p, Y5_OP_MAXPOOL, x, y, iv, 7, 0, 4.0 * batch * h * w * c);
}

extern "C" int y5_plan_add_maxpool_cascade(y5_plan* p, const void* x, int x_ld, void* y, int y_ld, int y_step, int levels, int batch,
                                           int h, int w, int c, int k) {
    Y5_REQUIRE(p, "plan_add_maxpool_cascade: null plan");
    const int iv[8] = {x_ld, y_ld, y_step, levels, batch, h, w, c};
    return add_pointwise(p, Y5_OP_MAXPOOL_CASCADE, x, y, iv, 8, k, 2.0 * batch * h * w * c * (1 + levels));
}

extern "C" int y5_plan_add_upsample2x(y5_plan* p, const void* x, int x_ld, void* y, int y_ld, int batch, int h, int w, int c, int mode) {
    Y5_REQUIRE(p, "plan_add_upsample2x: null plan");
    const int iv[7] = {x_ld, y_ld, batch, h, w, c, mode};
    return add_pointwise(p, Y5_OP_UPSAMPLE, x, y, iv, 7, 0, 10.0 * batch * h * w * c);
}

extern "C" int y5_plan_add_copy_channels(y5_plan* p, const void* x, int x_ld, void* y, int y_ld, int64_t pixels, int c) {
    Y5_REQUIRE(p, "plan_add_copy_channels: null plan");
    const int iv[3] = {x_ld, y_ld, c};
    return add_pointwise(p, Y5_OP_COPY, x, y, iv, 3, pixels, 4.0 * (double)pixels * c);
}

extern "C" int y5_plan_add_decode(y5_plan* p, const y5_decode_desc* d) {
    Y5_REQUIRE(p && d, "plan_add_decode: null argument");
    return guarded([&]() {
        Op op;
        op.kind = Y5_OP_DECODE;
        op.dec = *d;
        note_ptr(p, op, 0, d->head);
        note_ptr(p, op, 1, d->x_out);
        note_ptr(p, op, 2, d->z_out);
        // algorithmic bytes: read the head values once, write z (and the permuted raw copy x, when asked for) once, fp16
        op.bytes = (d->x_out ? 6.0 : 4.0) * d->batch * d->ny * d->nx * d->na * d->no;
        p->ops.push_back(op);
        return Y5_OK;
    });
}

extern "C" int y5_plan_add_decode_multi(y5_plan* p, const y5_decode_desc* levels, int n_levels) {
    Y5_REQUIRE(p && levels, "plan_add_decode_multi: null argument");
    Y5_REQUIRE(n_levels >= 1 && n_levels <= Y5_MAX_LEVELS, "plan_add_decode_multi: %d levels out of range 1..%d", n_levels, Y5_MAX_LEVELS);
    return guarded([&]() {
        Op op;
        op.kind = Y5_OP_DECODE_MULTI;
        op.i[0] = n_levels;
        for (int l = 0; l < n_levels; ++l) {
            const y5_decode_desc& d = levels[l];
            op.decm[l] = d;
            note_ptr(p, op, 3 * l + 0, d.head);
            note_ptr(p, op, 3 * l + 1, d.x_out);
            note_ptr(p, op, 3 * l + 2, d.z_out);
            op.bytes += (d.x_out ? 6.0 : 4.0) * d.batch * d.ny * d.nx * d.na * d.no;
        }
        p->ops.push_back(op);
        return Y5_OK;
    });
}

extern "C" int y5_plan_add_detect(y5_plan* p, const y5_conv_desc* conv, const y5_decode_desc* d, int variant) {
    Y5_REQUIRE(p && conv && d, "plan_add_detect: null argument");
    return guarded([&]() {
        Op op;
        op.kind = Y5_OP_DETECT;
        op.conv = *conv;
        op.dec = *d;
        op.i[0] = variant;
        note_ptr(p, op, 0, conv->x);
        note_ptr(p, op, 1, d->x_out);
        note_ptr(p, op, 2, d->z_out);
        const double m = (double)conv->batch * conv->h * conv->w_in, outs = (double)d->na * d->no;
        op.flops = 2.0 * m * conv->cout * conv->cin;
        // read the feature map and the weights once, write z (and the permuted raw copy, when asked for) once, fp16
        op.bytes = 2.0 * (m * conv->cin + (double)conv->cout * conv->cin + (d->x_out ? 2.0 : 1.0) * m * outs);
        p->ops.push_back(op);
        return Y5_OK;
    });
}

extern "C" int y5_plan_add_conv_cat2(y5_plan* p, const y5_conv_desc* d, const void* x2, int x2_ld, int cin2) {
    Y5_REQUIRE(p && d && x2, "plan_add_conv_cat2: null argument");
    return guarded([&]() {
        Op op;
        op.kind = Y5_OP_CONV_CAT2;
        op.conv = *d;
        op.conv2.x = x2;
        op.conv2.x_ld = x2_ld;
        op.conv2.cin = cin2;
        note_ptr(p, op, 0, d->x);
        note_ptr(p, op, 1, d->y);
        note_ptr(p, op, 2, d->res);
        note_ptr(p, op, 3, x2);
        const double m = (double)d->batch * d->ho * d->wo;
        op.flops = 2.0 * m * d->cout * (d->cin + cin2);
        op.bytes = 2.0 * (m / 4 * d->cin + m * cin2 + m * d->cout * (d->res ? 2 : 1) + (double)d->cout * (d->cin + cin2));
        p->ops.push_back(op);
        return Y5_OK;
    });
}

extern "C" int y5_plan_add_conv_chain(y5_plan* p, const y5_conv_desc* first, const y5_conv_desc* second, int keep_first, int variant) {
    Y5_REQUIRE(p && first && second, "plan_add_conv_chain: null argument");
    return guarded([&]() {
        Op op;
        op.kind = Y5_OP_CONV_CHAIN;
        op.conv = *first;
        op.conv2 = *second;
        op.i[0] = variant;
        op.i[1] = keep_first;
        note_ptr(p, op, 0, first->x);
        if (keep_first) note_ptr(p, op, 1, first->y);
        note_ptr(p, op, 2, first->res);
        note_ptr(p, op, 3, second->y);
        const double m = (double)first->batch * first->ho * first->wo;
        op.flops = 2.0 * m * first->cout * first->ksize * first->ksize * first->cin + 2.0 * m * second->cout * first->cout;
        op.bytes = 2.0 * ((double)first->batch * first->h * first->w_in * first->cin + m * first->cout * ((first->res ? 1 : 0) + (keep_first ? 1 : 0)) +
                          (double)first->cout * first->ksize * first->ksize * first->cin + m * second->cout + (double)second->cout * first->cout);
        p->ops.push_back(op);
        return Y5_OK;
    });
}

extern "C" int y5_plan_add_focus_conv2(y5_plan* p, const y5_conv_desc* f, const y5_conv_desc* g) {
    Y5_REQUIRE(p && f && g, "plan_add_focus_conv2: null argument");
    return guarded([&]() {
        Op op;
        op.kind = Y5_OP_FOCUS_CONV2;
        op.conv = *f;
        op.conv2 = *g;
        note_ptr(p, op, 0, f->x);
        note_ptr(p, op, 1, g->y);
        const double m0 = (double)f->batch * f->ho * f->wo, m1 = (double)g->batch * g->ho * g->wo;
        op.flops = 2.0 * m0 * f->cout * 36.0 * f->cin + 2.0 * m1 * g->cout * 9.0 * g->cin;
        op.bytes = 2.0 * ((double)f->batch * f->h * f->w_in * f->cin + m1 * g->cout + (double)f->cout * 36.0 * f->cin + (double)g->cout * 9.0 * g->cin);
        p->ops.push_back(op);
        return Y5_OK;
    });
}

extern "C" int y5_plan_add_csp(y5_plan* p, const y5_csp_desc* d) {
    Y5_REQUIRE(p && d, "plan_add_csp: null argument");
    return guarded([&]() {
        Op op;
        op.kind = Y5_OP_CSP;
        op.csp = *d;
        note_ptr(p, op, 0, d->x);
        note_ptr(p, op, 1, d->y);
        const double m = (double)d->batch * d->h * d->w;
        const double macs = (double)d->c1 * 2 * d->ch + (double)d->ch * d->ch * 11 + (double)2 * d->ch * d->c2;  // per pixel
        op.flops = 2.0 * m * macs;
        op.bytes = 2.0 * (m * d->c1 + m * d->c2 + macs);
        p->ops.push_back(op);
        return Y5_OK;
    });
}

extern "C" int y5_plan_mark_last(y5_plan* p, int flags) {
    Y5_REQUIRE(p && !p->ops.empty(), "plan_mark_last: no op recorded yet");
    Y5_REQUIRE((flags & ~(Y5_PLAN_SIDE | Y5_PLAN_JOIN)) == 0, "plan_mark_last: unknown flags 0x%x", flags);
    p->ops.back().flags |= flags;
    return Y5_OK;
}

extern "C" int y5_plan_num_ops(const y5_plan* p) { return p ? (int)p->ops.size() : 0; }

extern "C" int y5_plan_op_info(const y5_plan* p, int i, int* kind, double* flops, double* bytes) {
    Y5_REQUIRE(p && i >= 0 && i < (int)p->ops.size(), "plan_op_info: index %d out of range", i);
    if (kind) *kind = p->ops[(size_t)i].kind;
    if (flops) *flops = p->ops[(size_t)i].flops;
    if (bytes) *bytes = p->ops[(size_t)i].bytes;
    return Y5_OK;
}

namespace {

int ensure_side_stream(y5_plan* p) {
    if (p->side) return Y5_OK;
    Y5_HIP(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking));
    Y5_HIP(hipEventCreateWithFlags(&p->fork_ev, hipEventDisableTiming));
    Y5_HIP(hipEventCreateWithFlags(&p->join_ev, hipEventDisableTiming));
    return Y5_OK;
}

// main waits for everything issued to the side stream so far
int join_side(y5_plan* p, hipStream_t main) {
    Y5_HIP(hipEventRecord(p->join_ev, p->side));
    Y5_HIP(hipStreamWaitEvent(main, p->join_ev, 0));
    return Y5_OK;
}

}  // namespace

extern "C" int y5_plan_run(y5_plan* p, void* const* ext, int n_ext, y5_stream_t stream) {
    Y5_REQUIRE(p, "plan_run: null plan");
    hipStream_t s = static_cast<hipStream_t>(stream);
    return guarded([&]() {
        Op op;
        bool side_busy = false;
        for (size_t k = 0; k < p->ops.size(); ++k) {
            const Op* cur = &p->ops[k];
            if (!cur->fixes.empty()) {
                if (int rc = resolve(p, k, ext, n_ext, op)) return rc;
                cur = &op;
            }
            if ((cur->flags & Y5_PLAN_JOIN) && side_busy) {
                if (int rc = join_side(p, s)) return rc;
                side_busy = false;
            }
            if (cur->flags & Y5_PLAN_SIDE) {
                if (int rc = ensure_side_stream(p)) return rc;
                // the side stream starts this op once everything issued to the caller's stream so far is done
                Y5_HIP(hipEventRecord(p->fork_ev, s));
                Y5_HIP(hipStreamWaitEvent(p->side, p->fork_ev, 0));
                if (int rc = run_op(*cur, p->side)) return rc;
                side_busy = true;
            } else if (int rc = run_op(*cur, s)) {
                return rc;
            }
        }
        if (side_busy)
            if (int rc = join_side(p, s)) return rc;  // the caller's stream stands for the whole plan
        return (int)Y5_OK;
    });
}

extern "C" int y5_plan_profile(y5_plan* p, void* const* ext, int n_ext, y5_stream_t stream, float* ms, int n) {
    Y5_REQUIRE(p && ms, "plan_profile: null argument");
    Y5_REQUIRE(n >= (int)p->ops.size(), "plan_profile: ms holds %d entries, plan has %zu ops", n, p->ops.size());
    hipStream_t s = static_cast<hipStream_t>(stream);
    return guarded([&]() {
        const size_t nops = p->ops.size();
        std::vector<hipEvent_t> ev(nops + 1, nullptr);
        int rc = Y5_OK;
        for (size_t k = 0; k <= nops && rc == Y5_OK; ++k)
            if (hipEventCreate(&ev[k]) != hipSuccess) rc = y5_fail(Y5_EHIP, "plan_profile: hipEventCreate failed");
        Op op;
        if (rc == Y5_OK && hipEventRecord(ev[0], s) != hipSuccess) rc = y5_fail(Y5_EHIP, "plan_profile: hipEventRecord failed");
        for (size_t k = 0; k < nops && rc == Y5_OK; ++k) {
            rc = resolve(p, k, ext, n_ext, op);
            if (rc == Y5_OK) rc = run_op(op, s);
            if (rc == Y5_OK && hipEventRecord(ev[k + 1], s) != hipSuccess) rc = y5_fail(Y5_EHIP, "plan_profile: hipEventRecord failed");
        }
        if (rc == Y5_OK && hipStreamSynchronize(s) != hipSuccess) rc = y5_fail(Y5_EHIP, "plan_profile: hipStreamSynchronize failed");
        for (size_t k = 0; k < nops && rc == Y5_OK; ++k)
            if (hipEventElapsedTime(&ms[k], ev[k], ev[k + 1]) != hipSuccess) rc = y5_fail(Y5_EHIP, "plan_profile: hipEventElapsedTime failed");
        for (hipEvent_t e : ev)
            if (e) (void)hipEventDestroy(e);
        return rc;
    });
}
