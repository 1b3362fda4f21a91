/*
 * y5hip.h - C ABI of liby5hip.so: the MI355X (gfx950) kernels behind the YOLOv5-v2.0
 * inference hot path  conv forward -> Detect decode -> non_max_suppression.
 *
 * The reference has no FFI of its own for this path: it calls torch / torchvision tensor
 * operators from Python.  Every entry point below therefore cites the reference call site whose
 * arithmetic it replaces (paths relative to the reference tree).
 *
 * Conventions
 *   - plain C, no exceptions across the boundary; every function returns 0 on success or a
 *     negative Y5_E* code, with a human-readable message available from y5_last_error()
 *     (thread-local).
 *   - every pointer named in a descriptor is DEVICE memory owned by the caller, except where a
 *     comment says "host".  The library allocates no device memory and frees none.
 *   - every call is asynchronous on the hipStream_t passed as `stream` (NULL = the default
 *     stream) and uses the calling thread's current HIP device.
 *   - activations are NHWC fp16 ("pixel-major"): element (b, y, x, c) of a tensor with pixel
 *     pitch `ld` lives at  base + ((b*H + y)*W + x)*ld + c.  A tensor may be a channel slice of
 *     a wider buffer (ld > channels), which is how concatenations are formed in place.
 *     Channel counts and pitches are multiples of 8 (16-byte vectors); padding channels carry zeros.
 */
#ifndef Y5HIP_H
#define Y5HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define Y5_OK 0
#define Y5_EINVAL (-1)    /* bad argument / unsupported shape                     */
#define Y5_EHIP (-2)      /* a HIP runtime call failed (message has the HIP error) */
#define Y5_ENOMEM (-3)    /* host allocation failed                                */
#define Y5_EWORKSPACE (-4) /* caller-provided workspace too small                  */

typedef void* y5_stream_t; /* hipStream_t */

/* Library version (major*10000 + minor*100 + patch). */
int y5_version(void);
/* Message of the last failing call on this thread ("" if none). The pointer stays valid until
 * the next failing call on the same thread. */
const char* y5_last_error(void);
/* gcnArchName of the current device copied into buf (NUL-terminated, truncated to n). */
int y5_device_arch(char* buf, int n);

/* ------------------------------------------------------------------------------------------
 * Convolution as implicit GEMM on MFMA (fp16 in, fp32 accumulate, fp16 out).
 * Replaces: Conv.forward / Conv.fuseforward  models/common.py:26-30 (conv2d -> BatchNorm2d ->
 * LeakyReLU(0.1), or conv2d+bias -> LeakyReLU after Model.fuse models/yolo.py:155-163),
 * the plain 1x1 convs and the concat BatchNorm of BottleneckCSP  models/common.py:59-62,
 * the residual add of Bottleneck  models/common.py:43, and the biased 1x1 head conv of Detect
 * models/yolo.py:33.
 *
 * y[b,oy,ox,n] = post( sum_{r,q,c} x[b, oy*stride - pad + r, ox*stride - pad + q, c] * w[n, (r*k + q)*cin + c] )
 * post(v): v = fp16(v + bias[n])            (bias may be NULL)
 *          v = fp16(v*scale[n] + shift[n])  (only when scale != NULL; the folded BatchNorm)
 *          v = fp16(v > 0 ? v : slope*v)    (only when act == Y5_ACT_LEAKY)
 *          v = fp16(v + res[b,oy,ox,n])     (only when res != NULL)
 * The intermediate fp16 roundings are the ones the reference's tensor-by-tensor evaluation
 * performs in half precision.
 * ------------------------------------------------------------------------------------------ */
#define Y5_ACT_NONE 0
#define Y5_ACT_LEAKY 1
#define Y5_F16 0
#define Y5_BF16 1

typedef struct y5_conv_desc {
    const void* x;      /* input  (batch, h, w, cin) fp16, pixel pitch x_ld                         */
    const void* w;      /* weights fp16 [w_rows][kpad]; row n holds k = (r*ksize+q)*cin + c, zero padded */
    const float* bias;  /* [cout] fp32 or NULL                                                      */
    const float* scale; /* [cout] fp32 or NULL                                                      */
    const float* shift; /* [cout] fp32, required when scale != NULL                                 */
    const void* res;    /* residual (batch, ho, wo, cout) fp16, pixel pitch res_ld, or NULL         */
    void* y;            /* output (batch, ho, wo, cout) fp16, pixel pitch y_ld                      */
    int32_t batch, h, w_in, cin, x_ld;
    int32_t ho, wo, cout, y_ld, res_ld;
    int32_t ksize, stride, pad; /* ksize in {1,3}; stride in {1,2}; pad = ksize/2                  */
    int32_t kpad;               /* >= ksize*ksize*cin, multiple of 64                               */
    int32_t w_rows;             /* >= cout, multiple of 128 (256 for the variants with 256-channel tiles) */
    int32_t act;                /* Y5_ACT_*                                                         */
    float slope;                /* LeakyReLU negative slope (0.1 in the reference)                  */
    int32_t variant;            /* 0 = tile configuration chosen by rule; 1..y5_conv_num_variants() = explicit   */
    int32_t dtype;              /* Y5_F16 (0) or Y5_BF16: element type of x, w, res and y.  bf16 (the reference's modules
                                 * under .bfloat16(): same operations, bfloat16 roundings) is served by y5_conv_f16 with the
                                 * im2col kernel family and the phased kernels: variant 0 or ids 1..y5_conv_num_variants_bf16();
                                 * every other entry point that takes this descriptor wants Y5_F16 */
} y5_conv_desc;

int y5_conv_f16(const y5_conv_desc* d, y5_stream_t stream);
/* Number of compiled kernel variants (tile configurations of the im2col kernel, then the halo, weight-stationary and
 * loader-wave kernel families), for autotuning the `variant` field: time each id on the real shape and keep the
 * fastest; an id that does not apply to the shape returns Y5_EINVAL. */
int y5_conv_num_variants(void);
/* Number of kernel variants compiled for Y5_BF16 (ids 1..n of the variant field when dtype is Y5_BF16): the im2col tile
 * configurations, then the phased kernels. */
int y5_conv_num_variants_bf16(void);
/* Number of phased implicit-GEMM kernels (the compute-bound layers of models/common.py:18-30 at the widths
 * models/yolo.py:195 produces for yolov5m / l / x): the LAST ids of y5_conv_num_variants() and of
 * y5_conv_num_variants_bf16(). */
int y5_conv_num_variants_phased(void);

/* Two convolutions in one kernel: `first` (3x3, or 1x1 stride 1; all post-ops of y5_conv_f16 including the residual)
 * followed by `second`, a 1x1 stride-1 convolution of first's output (bias / scale+shift / activation, no residual):
 *     second.y = post2(W2 . first_result)
 * with first_result rounded to fp16 exactly as y5_conv_f16 would store it.  Replaces two consecutive Conv.forward /
 * nn.Conv2d calls (models/common.py:26, :43, :61: the last Bottleneck's cv2 followed by BottleneckCSP.cv3) without the
 * intermediate tensor: first.y is written only if keep_first != 0 (it may be NULL otherwise); second.x is ignored.
 * Both channel counts must fit one tile of the chosen variant (at most 256; Y5_EINVAL otherwise) and both weight
 * tensors must be packed to that many rows.  variant: 0 or 1..y5_conv_chain_num_variants().  Same values as the two
 * separate calls up to the fp32 summation order.
 * second.cin may be smaller than first.cout: second then reads the LEADING second.cin channels of first's result
 * (BottleneckCSP.cv1 and cv2 recorded as one convolution, followed by the first Bottleneck's cv1, which reads the cv1
 * half: models/common.py:59-60, :41).  Only the last variant but one, a persistent 1x1 kernel compiled for
 * 256 -> 256 with 128 -> 128 and 128 -> 128 with 64 -> 64 channels, takes such a pair (variant 0 selects it). */
int y5_conv_chain_f16(const y5_conv_desc* first, const y5_conv_desc* second, int keep_first, int variant, y5_stream_t stream);
int y5_conv_chain_num_variants(void);

/* BottleneckCSP with ONE Bottleneck as one kernel: every intermediate tensor of the block stays on chip.
 * Replaces: BottleneckCSP.forward  models/common.py:59-62 with n = 1, i.e. the five convolutions
 *     y1 = cv3(m[0](cv1(x))),  m[0](t) = t + m[0].cv2(m[0].cv1(t))  (Bottleneck.forward models/common.py:43; without the
 *     y2 = cv2(x)                                                      addition when shortcut == 0)
 *     out = cv4(act(bn(cat(y1, y2))))
 * Each stage carries the post-ops of y5_conv_f16 (bias / scale+shift / LeakyReLU, each rounded to fp16), so the result
 * equals the layer-by-layer evaluation up to the fp32 summation order:
 *     cv1 with its own BatchNorm or bias;   cv2 with the second half of the concat BatchNorm;
 *     m_cv1, m_cv2 = the Bottleneck's convolutions (m_cv2 is 3x3, k = (r*3+q)*ch + c);
 *     cv3 with the first half of the concat BatchNorm;   cv4 reads [y1 | y2].
 * Compiled for c1 = 64, ch = 32, c2 = 64 (yolov5s layer 2), h % 8 == 0, w % 32 == 0; Y5_EINVAL otherwise.  Every stage
 * is followed by LeakyReLU(slope).  Two post-op patterns are compiled: the Model.fuse()d block (bias only on cv1, m_cv1,
 * m_cv2, cv4; scale + shift only on cv2, cv3) and the block with live BatchNorms (scale + shift only everywhere). */
typedef struct y5_csp_stage {
    const void* w;      /* fp16 [rows][w_ld], row n = output channel n, zero padded */
    const float* bias;  /* fp32 per output channel, or NULL */
    const float* scale; /* fp32 per output channel, or NULL */
    const float* shift; /* required when scale != NULL */
    int w_ld;           /* row pitch in elements */
    int reserved;
} y5_csp_stage;
typedef struct y5_csp_desc {
    const void* x; /* (batch, h, w, c1) fp16, pixel pitch x_ld */
    void* y;       /* (batch, h, w, c2) fp16, pixel pitch y_ld */
    int batch, h, w;
    int c1, x_ld, ch, c2, y_ld;
    int shortcut;
    float slope;
    y5_csp_stage cv1, cv2, m_cv1, m_cv2, cv3, cv4;
} y5_csp_desc;
int y5_csp_f16(const y5_csp_desc* d, y5_stream_t stream);

/* A 1x1 stride-1 convolution whose input is the channel concatenation [nearest-2x-upsampled x | x2], without building
 * the upsampled tensor or the concatenation.
 * Replaces: nn.Upsample(None, 2, 'nearest') -> Concat -> the 1x1 convolutions that read it (models/yolov5s.yaml:30-32 and
 * :34-36: layers 11-13 and 15-17; BottleneckCSP.cv1 / cv2, models/common.py:59-61).
 * Descriptor as for y5_conv_f16 with ksize = 1, stride = 1, pad = 0 and these meanings: h = ho, w_in = wo = the OUTPUT
 * size (even); x = the half-resolution tensor (batch, ho/2, wo/2, cin) with pitch x_ld, cin a multiple of 64; x2 =
 * (batch, ho, wo, cin2) with pitch x2_ld.  A weight row holds cin + cin2 values, x's channels first (kpad >= cin + cin2).
 * Output pixel (oy, ox) reads x at (oy/2, ox/2) and x2 at (oy, ox).  variant: 0 or 1..y5_conv_cat2_num_variants(); the
 * last one is a persistent kernel with the weights in registers, compiled for 128 + 128 -> 128 and 256 + 256 -> 256
 * channels without a residual (Y5_EINVAL for other shapes).
 * Same values as upsampling, concatenating and calling y5_conv_f16. */
int y5_conv_cat2_f16(const y5_conv_desc* d, const void* x2, int x2_ld, int cin2, y5_stream_t stream);
int y5_conv_cat2_num_variants(void);

/* Focus followed by its 3x3 stride-1 pad-1 convolution, straight from the NCHW image, without materialising the
 * space-to-depth tensor.  Replaces: Focus.forward  models/common.py:85-86 (four strided slices + cat + self.conv).
 * Same descriptor and post-ops as y5_conv_f16 with these meanings: x = image (batch, cin = 3, h, w_in) fp16 NCHW
 * (x_ld unused), ho = h/2, wo = w_in/2 (h, w_in even); w = the convolution's weights packed as y5_conv_f16 wants
 * them for the 16-channel (12 + 4 zero) space-to-depth input, i.e. k = (r*3 + q)*16 + (dy + 2*dx)*3 + c for image
 * channel c at row parity dy, column parity dx; cout <= 96; res must be NULL; variant is ignored. */
int y5_focus_conv_f16(const y5_conv_desc* d, y5_stream_t stream);

/* Focus + its convolution AND the 3x3 stride-2 convolution that follows it, as one kernel: the tensor between the two
 * layers is never written.  Replaces: Focus.forward models/common.py:85-86 followed by Conv.forward / fuseforward :26, :29
 * (models/yolov5s.yaml backbone rows 0 and 1).  focus: as for y5_focus_conv_f16 (fp16 NCHW image, 32 filters; y ignored);
 * conv: the second layer's descriptor (3x3, stride 2, 32 -> 64 channels, input = the first layer's output; x ignored) with
 * conv.ho % 8 == 0 and conv.wo % 16 == 0.  Both layers either bias-only (fused model) or scale + shift only (BatchNorm).
 * Values: for bias-only layers bit-identical to y5_focus_conv_f16 followed by y5_conv_f16's persistent stride-2 variant
 * (same summation order); with scale + shift the affine step is one fused multiply-add (isolated values one fp16 step apart).
 * focus->variant chooses the schedule, not the arithmetic (both give the same bits): 0 = tile by tile, 1 = the second layer
 * of a tile beside the first layer of the next tile (one barrier per tile); other values are Y5_EINVAL.  conv->variant is
 * ignored. */
int y5_focus_conv2_f16(const y5_conv_desc* focus, const y5_conv_desc* conv, y5_stream_t stream);

/* Image pre-processing on the device, the step before the path: img.half() followed by img /= 255.0
 * (test.py:152-154, detect.py:57-59): y[i] = fp16(float(x[i]) / 255.0f) for n uint8 values (any layout). */
int y5_normalize_u8_f16(const void* x_u8, void* y_f16, int64_t n, y5_stream_t stream);
/* y5_focus_conv_f16 with that pre-processing folded in: x is the uint8 NCHW image as the data loader delivers it
 * (utils/datasets.py:140-141 after .to(device)); results are bit-identical to normalising first. */
int y5_focus_conv_u8(const y5_conv_desc* d, y5_stream_t stream);
/* The same from the decoder's own layout: x is uint8 (batch, h, w_in, 3), channels interleaved, i.e. the image before the
 * loader's `img[:, :, ::-1].transpose(2, 0, 1)` + np.ascontiguousarray (utils/datasets.py:140-141).  input = Y5_IN_U8_HWC
 * takes the three bytes of a pixel as the model's channels 0, 1, 2; Y5_IN_U8_HWC_BGR reverses them (the `::-1`: OpenCV's
 * B, G, R become R, G, B).  Swap, transpose and the division by 255 happen in the kernel's fetch; results are bit-identical
 * to running the loader's statements first. */
#define Y5_IN_F16 0
#define Y5_IN_U8 1
#define Y5_IN_U8_HWC 2
#define Y5_IN_U8_HWC_BGR 3
int y5_focus_conv_u8_hwc(const y5_conv_desc* d, int input, y5_stream_t stream);

/* A first-layer 3x3 stride-1 pad-1 convolution straight from the 3-channel NCHW image, for the graphs that open with a
 * plain Conv instead of Focus (models/yolov5s_ori.yaml layer 0; Conv.forward models/common.py:26 on the loader's image).
 * Descriptor as for y5_focus_conv_f16 (x = image (batch, 3, h, w_in) NCHW, x_ld ignored, weights packed as for an
 * 8-channel NHWC input: k = (r*3+q)*8 + c, kpad >= 72) with ho = h, wo = w_in and cout <= 64.  _u8: the uint8 image,
 * divided by 255 on the way in like y5_focus_conv_u8. */
int y5_stem_conv_f16(const y5_conv_desc* d, y5_stream_t stream);
int y5_stem_conv_u8(const y5_conv_desc* d, y5_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Input packing: NCHW fp16 image batch -> NHWC fp16, optionally with the Focus space-to-depth
 * gather. Replaces: Focus.forward's four strided slices + cat  models/common.py:85-86
 * (output channel q*c + ci, q = (row parity, col parity) in the order (0,0),(1,0),(0,1),(1,1)).
 * focus = 0: y is (batch, h, w, y_ld), channels [c, y_ld) zero.
 * focus = 1: y is (batch, h/2, w/2, y_ld), channels [4c, y_ld) zero; h and w must be even.
 * ------------------------------------------------------------------------------------------ */
int y5_pack_image_f16(const void* x_nchw, void* y_nhwc, int batch, int c, int h, int w, int focus, int y_ld,
                      y5_stream_t stream);

/* k x k max pooling, stride 1, padding k/2 with -inf (k odd, <= 13), on a channel slice.
 * Replaces: nn.MaxPool2d(k, 1, k//2) inside SPP  models/common.py:72,76.  x and y may be
 * different channel slices of the same buffer (the SPP concat) but must not overlap. */
int y5_maxpool_f16(const void* x, int x_ld, void* y, int y_ld, int batch, int h, int w, int c, int k,
                   y5_stream_t stream);

/* All pooling levels of SPP in one launch: level l (0-based) is the k x k stride-1 max pool applied l+1 times,
 * i.e. a single pool of size l*(k-1)+k because max pooling composes exactly; k = 5 with 3 levels gives the
 * reference's (5, 9, 13).  Replaces: the nn.MaxPool2d list of SPP  models/common.py:72,76.  Level l is written
 * to channels [l*y_step, l*y_step + c) of y.  The h*w plane must fit in LDS (h*w*32 bytes <= 64 KiB). */
int y5_maxpool_cascade_f16(const void* x, int x_ld, void* y, int y_ld, int y_step, int levels, int batch, int h, int w,
                           int c, int k, y5_stream_t stream);

/* 2x spatial upsampling of a channel slice into another (usually a concat) buffer.
 * Replaces: nn.Upsample(None, 2, 'nearest' | 'bilinear') + Concat  models/yolov5s.yaml:30-31,
 * models/yolov5s_ori.yaml:79-80.  mode 0 = nearest, 1 = bilinear with align_corners=False. */
int y5_upsample2x_f16(const void* x, int x_ld, void* y, int y_ld, int batch, int h, int w, int c, int mode,
                      y5_stream_t stream);

/* The three pointwise kernels above that do arithmetic on the values (a maximum, a blend) for either 16-bit element type:
 * dtype = Y5_F16 gives exactly the _f16 entry points, Y5_BF16 the same operations on bfloat16 tensors.  (Packing, nearest
 * upsampling and channel copies only move 16-bit values: their _f16 entry points serve both types.) */
int y5_maxpool_x16(const void* x, int x_ld, void* y, int y_ld, int batch, int h, int w, int c, int k, int dtype, y5_stream_t stream);
int y5_maxpool_cascade_x16(const void* x, int x_ld, void* y, int y_ld, int y_step, int levels, int batch, int h, int w, int c, int k,
                           int dtype, y5_stream_t stream);
int y5_upsample2x_x16(const void* x, int x_ld, void* y, int y_ld, int batch, int h, int w, int c, int mode, int dtype,
                      y5_stream_t stream);

/* Copy a channel slice (batch*h*w pixels, c channels) between pitched buffers.
 * Replaces: Concat.forward  models/common.py:95-96 for operands that cannot be produced in place. */
int y5_copy_channels_f16(const void* x, int x_ld, void* y, int y_ld, int64_t pixels, int c, y5_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * Train-mode BatchNorm2d on NHWC fp16 tensors: the statistics-recalibration ("adaptive BN") forward.
 * Replaces: the nn.BatchNorm2d calls of Conv.forward (models/common.py:27) and BottleneckCSP.forward
 * (models/common.py:62) while the model is in train() mode under torch.no_grad(), as
 * Training.adaptive_BN runs it (filter_pruning/fp_train.py:231-240): normalise with the batch's own mean and
 * biased variance, move running_mean / running_var by `momentum` (running_var with the unbiased variance).
 * A convolution's raw output is produced by y5_conv_f16 without post-ops; then, per BatchNorm:
 *
 * y5_bn_stats_rows:    number of rows R(pixels, c) of partial sums y5_bn_stats_f16 writes for such a tensor
 *                     (<= 512; 0 for arguments y5_bn_stats_f16 rejects).
 * y5_bn_stats_f16:    partials[r][0..c) = per-channel sums, partials[r][c..2c) = per-channel sums of squares of the
 *                     r-th share of the `pixels` pixels, r < R (fp32, R*2c values, overwritten; no atomics, so
 *                     repeated calls give identical bits).  c a multiple of 8, <= 2048.
 * y5_bn_finalize_f32: adds the `rows` partial rows in fp64, then scale[i] = gamma[i] / sqrt(var_b[i] + eps),
 *                     shift[i] = beta[i] - mean[i]*scale[i] for i < c_valid, 0 for the padding channels
 *                     c_valid <= i < c; running statistics (fp32, may be NULL) updated in place.  gamma / beta may
 *                     be NULL (1 / 0).  count = values per channel (= pixels).
 * y5_affine_act_f16:  y = fp16(act(x*scale[c] + shift[c]) + res), evaluated in fp32 and rounded once; act is
 *                     Y5_ACT_NONE or Y5_ACT_LEAKY(slope); res may be NULL (Bottleneck's shortcut,
 *                     models/common.py:43).  y may alias x.
 * ---------------------------------------------------------------------------------------------- */
/* y5_conv_f16 that also ADDS the per-channel sums of the outputs it stores (the fp16 values, after every post-op) into
 * partials (fp32 [rows][2*stats_c], zeroed by the caller before the first convolution that adds to it): output channel n
 * goes to columns stats_off + n (sum) and stats_c + stats_off + n (sum of squares) of row (pixel tile index % rows).
 * y5_bn_finalize_f32(partials, rows, ...) then replaces y5_bn_stats_f16 + its pass over the tensor.  Only variant 0 and the
 * im2col kernel family (the first variant ids) accumulate; for the other kernel families the call returns Y5_EINVAL.
 * fp32 atomics: sums can differ in the last bits between runs. */
int y5_conv_stats_f16(const y5_conv_desc* d, float* partials, int rows, int stats_c, int stats_off, y5_stream_t stream);
int y5_bn_stats_rows(int64_t pixels, int c);
int y5_bn_stats_f16(const void* x, int x_ld, int64_t pixels, int c, float* partials, y5_stream_t stream);
int y5_bn_finalize_f32(const float* partials, int rows, int64_t count, int c, int c_valid, const float* gamma, const float* beta,
                       float eps, float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                       y5_stream_t stream);
int y5_affine_act_f16(const void* x, int x_ld, void* y, int y_ld, const void* res, int res_ld, int64_t pixels, int c,
                      const float* scale, const float* shift, int act, float slope, y5_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Detect box decode for one pyramid level.
 * Replaces: Detect.forward  models/yolo.py:34-46 (view/permute/contiguous, sigmoid, grid/anchor
 * decode, per-level part of cat(z, 1)) and _make_grid  models/yolo.py:48-51.
 * head: (batch, ny, nx, head_ld) fp16, channel a*no + o.
 * x_out: (batch, na, ny, nx, no) fp16 raw head values (the reference's x[i]); may be NULL.
 * z_out: (batch, z_rows_total, no) fp16; this level writes rows z_row0 + a*ny*nx + y*nx + x.
 * Rounding order (half mode): s = fp16(sigmoid(v)); xy = fp16((fp32(fp16(2s - 0.5)) + grid) * stride);
 * wh = fp16(fp16((2s)^2) * anchor); everything else s.
 * ------------------------------------------------------------------------------------------ */
#define Y5_MAX_ANCHORS 8
typedef struct y5_decode_desc {
    const void* head;
    void* x_out;
    void* z_out;
    int32_t head_ld;
    int32_t batch, ny, nx, na, no;
    int32_t z_rows_total, z_row0;
    float stride;
    float anchor_wh[2 * Y5_MAX_ANCHORS]; /* pixel anchors (w,h) per anchor as stored in Detect.anchor_grid */
    int32_t dtype;                       /* Y5_F16 or Y5_BF16: element type of head, x_out and z_out; the rounding order above
                                          * then holds with bfloat16 in place of fp16 (Detect.forward under .bfloat16()) */
} y5_decode_desc;

int y5_detect_decode_f16(const y5_decode_desc* d, y5_stream_t stream);
/* The same for n_levels (<= Y5_MAX_LEVELS) pyramid levels in ONE launch: the loop `for i in range(self.nl)` of
 * Detect.forward  models/yolo.py:31-46.  Results are identical to n_levels separate calls. */
#define Y5_MAX_LEVELS 4
int y5_detect_decode_multi_f16(const y5_decode_desc* levels, int n_levels, y5_stream_t stream);

/* One Detect level in one kernel: the level's 1x1 convolution self.m[i] (bias, no activation) with the decode above as
 * its epilogue.  Replaces: Detect.forward  models/yolo.py:32-46 for level i (conv, view/permute/contiguous, sigmoid, box
 * decode, view into z) without the NHWC head tensor in between.  conv: ksize 1, stride 1, cout = na*no padded to a
 * multiple of 8 (<= 256), weights packed to 256 rows, y unused; d: as for y5_detect_decode_f16 with head unused.  Needs
 * the 16-byte store alignment of the decode kernel's fast path (ny*nx*no, z_rows_total*no and z_row0*no multiples of 8),
 * else Y5_EINVAL: use the two separate calls then.  variant: 0 or 1..y5_detect_num_variants() (tile configurations, for
 * autotuning like y5_conv_desc.variant).  Same values as the two-kernel path up to the fp32 summation order of the
 * convolution. */
int y5_detect_f16(const y5_conv_desc* conv, const y5_decode_desc* d, int variant, y5_stream_t stream);
int y5_detect_num_variants(void);

/* y[i] = fp16(sigmoid(x[i])) for n fp16 values, the elementwise op of Detect.forward (x[i].sigmoid(), models/yolo.py:41)
 * exactly as the decode kernel evaluates it.  libm = 0: the kernel's exp2/reciprocal form; libm = 1: 1/(1+expf(-x)) with
 * IEEE division.  Both must agree on all 65536 inputs; the entry point exists so that this can be tested. */
int y5_sigmoid_f16(const void* x, void* y, int64_t n, int libm, y5_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Batched non-maximum suppression.
 * Replaces: non_max_suppression  utils/utils.py:564-645 for merge=False (confidence filter,
 * conf = obj*cls, xywh2xyxy utils/utils.py:183-190, multi-label candidate expansion, optional
 * class filter, per-class 4096-pixel box offset, torchvision.ops.boxes.nms call utils/utils.py:627,
 * 300-detection cap).  torchvision's greedy NMS: visit boxes by descending score (stable: lower
 * original index first among equal scores); a visited box is kept unless an earlier KEPT box has
 * inter/(area_i + area_j - inter) > iou_thres, all in fp32 and compared in double.
 *
 * pred: (batch, n_rows, no) rows [cx, cy, w, h, obj, cls_0..]; pred_dtype 0 = fp16, 1 = fp32, 2 = bf16
 * (every value is widened to fp32 before any arithmetic, as the reference does for fp16: utils/utils.py:580-581).
 * det: (batch, max_det, 6) fp32 rows [x1, y1, x2, y2, conf, cls], first count[b] rows valid, rest zero.
 * kept_index (optional): (batch, max_det) int32, anchor_row*nc + cls of each kept detection.
 * n_candidates (optional): (batch) int32, candidates that passed the confidence filter (uncapped).
 * If n_candidates[b] > max_candidates the image was truncated to an arbitrary subset and the
 * caller must retry with a larger capacity.
 * ------------------------------------------------------------------------------------------ */
#define Y5_NMS_AGNOSTIC 1 /* one class for the IoU test (the `agnostic` argument, utils/utils.py:625) */
#define Y5_NMS_MERGE 2    /* the merge step after the selection (the `merge` argument, utils/utils.py:630-639): in an image with
                           * 1 < n < 3000 candidates every kept box becomes the score-weighted mean of all candidates that
                           * overlap it by more than iou_thres, and kept boxes no second candidate overlaps are dropped; count
                           * and kept_index follow.  Sums in fp64, rounded once: equal to the reference's fp32 matrix product
                           * up to its summation order (a few ulp) */
typedef struct y5_nms_desc {
    const void* pred;
    const uint8_t* class_keep; /* [nc] 1 = class allowed, or NULL = all (the `classes` argument) */
    void* workspace;
    float* det;
    int32_t* count;
    int32_t* kept_index;
    int32_t* n_candidates;
    size_t workspace_bytes;
    double iou_thres;
    float conf_thres;
    int32_t pred_dtype;
    int32_t batch, n_rows, no;
    int32_t agnostic;       /* flags: Y5_NMS_AGNOSTIC | Y5_NMS_MERGE */
    int32_t max_candidates; /* per-image candidate capacity */
    int32_t max_det;        /* <= 1024; 300 in the reference */
} y5_nms_desc;

size_t y5_nms_workspace_bytes(int batch, int max_candidates);
int y5_nms_batched(const y5_nms_desc* d, y5_stream_t stream);

/* y5_nms_batched for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  Same descriptor, same results bit for bit;
 * d->workspace and d->workspace_bytes are ignored: counters and sort keys live in a device array inside the library,
 * Y5_NMS_NOALLOC_BYTES per GPU the library runs on, held until the process ends.  The call allocates and frees
 * nothing.  y5_nms_workspace_bytes(batch, max_candidates) must not exceed Y5_NMS_NOALLOC_BYTES (batch 32 with 65536
 * candidates per image just fits), else Y5_EWORKSPACE.  All calls on one GPU share the array: calls
 * that overlap in time on one GPU (different streams or threads) race; calls on one stream are ordered and safe. */
#define Y5_NMS_NOALLOC_BYTES ((size_t)(16u << 20) + 4096u) /* 2 Mi sort keys (batch * pow2(max_candidates)) + counters of up to 1024 images */
int y5_nms_batched_noalloc(const y5_nms_desc* d, y5_stream_t stream);

/* Prediction <-> label matching of the evaluation loop, for a whole batch, without a host round trip.
 * Replaces: test.py:199 (clip_coords(pred, (height, width)), done in place on det) and test.py:220-247 (per class:
 * box_iou(pred, tbox).max(1), predictions in confidence order claim their best label once, correct = iou > iouv).
 * det (batch, max_det, row_floats >= 6) fp32 rows xyxy, conf, cls with count[b] valid rows, as y5_nms_batched wrote
 * them; targets (n_targets, 6) fp32 rows {image index, class, x, y, w, h normalised} (the data loader's `targets`);
 * iouv (niou <= 16) fp32; correct (batch, max_det, niou) bytes, 0/1, rows past count[b] zero; overflow (batch) int32 or
 * NULL: set to 1 for an image with more than 2048 labels (the excess is ignored). */
int y5_match_predictions_f32(float* det, const int32_t* count, const float* targets, int n_targets, const float* iouv, int niou,
                             uint8_t* correct, int32_t* overflow, int batch, int max_det, int row_floats, float height, float width,
                             y5_stream_t stream);

/* Rescale the kept boxes from the network's input size to each original image and clip them, in place, without a host
 * round trip.  Replaces: scale_coords + clip_coords  utils/utils.py:193-214 (called per image at detect.py:93 and
 * test.py:206,231), applied to the padded block that y5_nms_batched wrote: det (batch, max_det, row_floats) fp32 with
 * xyxy in floats 0..3, count[b] valid rows per image; params (batch, 5) fp32 = {pad_x, pad_y, gain, img0_w, img0_h} as
 * the reference computes them (or takes them from ratio_pad).  do_round != 0 adds detect.py's .round(). */
int y5_scale_coords_f32(float* det, const int32_t* count, const float* params, int batch, int max_det, int row_floats,
                        int do_round, y5_stream_t stream);

/* Greedy NMS of one box set, the drop-in for torchvision.ops.boxes.nms(boxes, scores, iou_thres)
 * (call site utils/utils.py:627; also used by the merge=True branch utils/utils.py:630-639).
 * boxes (n,4) fp32 corner boxes, scores (n) fp32.  keep (max_keep) int64 receives the kept indices
 * in descending score order, keep_count[0] their number (<= max_keep).
 * workspace: y5_nms_workspace_bytes(1, n) bytes. */
int y5_nms_boxes_f32(const float* boxes, const float* scores, int n, double iou_thres, int64_t* keep,
                     int32_t* keep_count, int max_keep, void* workspace, size_t workspace_bytes,
                     y5_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Launch plans: a host-side list of the calls above, replayed with one C call per forward.
 * Replaces: the Python module walk of Model.forward_once  models/yolo.py:107-130.
 * Pointers inside the recorded descriptors are used as recorded, except pointers that were
 * registered as "external slot k": those are replaced at run time by ext[k] + recorded offset
 * (network input, per-call output tensors).
 * ------------------------------------------------------------------------------------------ */
typedef struct y5_plan y5_plan;

int y5_plan_create(y5_plan** out);
void y5_plan_destroy(y5_plan* p);
/* Declare that device address `base` stands for external slot `slot` (0..15) while recording. */
int y5_plan_bind_slot(y5_plan* p, int slot, const void* base, size_t bytes);
/* Element type (Y5_F16 / Y5_BF16) of the pooling and upsampling ops recorded after this call; convolutions and decodes
 * carry theirs in their descriptors.  A new plan records Y5_F16. */
int y5_plan_set_dtype(y5_plan* p, int dtype);
int y5_plan_add_conv(y5_plan* p, const y5_conv_desc* d);
int y5_plan_add_focus_conv(y5_plan* p, const y5_conv_desc* d);
int y5_plan_add_focus_conv_u8(y5_plan* p, const y5_conv_desc* d);
int y5_plan_add_stem_conv(y5_plan* p, const y5_conv_desc* d);
int y5_plan_add_stem_conv_u8(y5_plan* p, const y5_conv_desc* d);
int y5_plan_add_normalize_u8(y5_plan* p, const void* x_u8, void* y_f16, int64_t n);
int y5_plan_add_pack_image(y5_plan* p, const void* x_nchw, void* y_nhwc, int batch, int c, int h, int w, int focus,
                           int y_ld);
int y5_plan_add_maxpool(y5_plan* p, const void* x, int x_ld, void* y, int y_ld, int batch, int h, int w, int c,
                        int k);
int y5_plan_add_maxpool_cascade(y5_plan* p, const void* x, int x_ld, void* y, int y_ld, int y_step, int levels, int batch,
                                int h, int w, int c, int k);
int y5_plan_add_upsample2x(y5_plan* p, const void* x, int x_ld, void* y, int y_ld, int batch, int h, int w, int c,
                           int mode);
int y5_plan_add_copy_channels(y5_plan* p, const void* x, int x_ld, void* y, int y_ld, int64_t pixels, int c);
int y5_plan_add_decode(y5_plan* p, const y5_decode_desc* d);
int y5_plan_add_decode_multi(y5_plan* p, const y5_decode_desc* levels, int n_levels);
int y5_plan_add_detect(y5_plan* p, const y5_conv_desc* conv, const y5_decode_desc* d, int variant);
int y5_plan_add_conv_chain(y5_plan* p, const y5_conv_desc* first, const y5_conv_desc* second, int keep_first, int variant);
int y5_plan_add_conv_cat2(y5_plan* p, const y5_conv_desc* d, const void* x2, int x2_ld, int cin2);
int y5_plan_add_csp(y5_plan* p, const y5_csp_desc* d);
int y5_plan_add_focus_conv_u8_hwc(y5_plan* p, const y5_conv_desc* d, int input);
int y5_plan_add_focus_conv2(y5_plan* p, const y5_conv_desc* focus, const y5_conv_desc* conv);
/* Mark the op recorded last.  Y5_PLAN_SIDE: y5_plan_run launches it on a second stream owned by the plan, ordered
 * after everything issued before it on the caller's stream -- for an op whose result is needed only much later (the
 * Detect convs of the finer pyramid levels, which can run beside the rest of the head).  Y5_PLAN_JOIN: the caller's
 * stream waits for the side stream before this op.  The end of the plan always joins.  One y5_plan_run at a time per
 * plan.  y5_plan_profile ignores the marks (everything in order on one stream). */
#define Y5_PLAN_SIDE 1
#define Y5_PLAN_JOIN 2
int y5_plan_mark_last(y5_plan* p, int flags);
int y5_plan_num_ops(const y5_plan* p);
/* Replay all ops on `stream`. ext (host array of n_ext device pointers) supplies the external slots. */
int y5_plan_run(y5_plan* p, void* const* ext, int n_ext, y5_stream_t stream);
/* Replay with a hipEvent pair around every op; ms (host, n >= num_ops floats) receives per-op
 * device time. Synchronises the stream. For measurement only. */
int y5_plan_profile(y5_plan* p, void* const* ext, int n_ext, y5_stream_t stream, float* ms, int n);
/* Kind (Y5_OP_*), algorithmic FLOPs and algorithmic bytes of op i (host-side bookkeeping). */
#define Y5_OP_CONV 1
#define Y5_OP_PACK 2
#define Y5_OP_MAXPOOL 3
#define Y5_OP_UPSAMPLE 4
#define Y5_OP_COPY 5
#define Y5_OP_DECODE 6
#define Y5_OP_MAXPOOL_CASCADE 7
#define Y5_OP_FOCUS_CONV 8
#define Y5_OP_DECODE_MULTI 9
#define Y5_OP_FOCUS_CONV_U8 10
#define Y5_OP_NORMALIZE_U8 11
#define Y5_OP_DETECT 12
#define Y5_OP_CONV_CHAIN 13
#define Y5_OP_STEM_CONV 14
#define Y5_OP_STEM_CONV_U8 15
#define Y5_OP_CONV_CAT2 16
#define Y5_OP_CSP 17
#define Y5_OP_FOCUS_CONV_HWC 18
#define Y5_OP_FOCUS_CONV2 19
int y5_plan_op_info(const y5_plan* p, int i, int* kind, double* flops, double* bytes);

#ifdef __cplusplus
}
#endif
#endif /* Y5HIP_H */
