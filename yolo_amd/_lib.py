"""ctypes binding of liby5hip.so (the C ABI declared in include/y5hip.h).

The library is built in-tree by `make -C yolo_amd/csrc` (or __graft_entry__.build()).  There is no
fallback: if it is missing, every GPU entry point of this package raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'liby5hip.so')

Y5_MAX_ANCHORS = 8
Y5_MAX_LEVELS = 4
NMS_NOALLOC_BYTES = (16 << 20) + 4096  # Y5_NMS_NOALLOC_BYTES
PLAN_SIDE, PLAN_JOIN = 1, 2
NMS_AGNOSTIC, NMS_MERGE = 1, 2  # flags of y5_nms_desc.agnostic
ACT_NONE, ACT_LEAKY = 0, 1
F16, BF16 = 0, 1  # y5_conv_desc.dtype / y5_decode_desc.dtype
IN_F16, IN_U8, IN_U8_HWC, IN_U8_HWC_BGR = 0, 1, 2, 3  # input layouts of y5_focus_conv_u8_hwc
OK, EINVAL, EHIP, ENOMEM, EWORKSPACE = 0, -1, -2, -3, -4  # y5_status in include/y5hip.h
OP_NAMES = {19: 'focus_conv', 18: 'focus_conv', 17: 'conv', 16: 'conv', 14: 'focus_conv', 15: 'focus_conv', 13: 'conv', 12: 'detect', 8: 'focus_conv', 9: 'decode', 10: 'focus_conv', 11: 'normalize', 1: 'conv', 2: 'pack', 3: 'maxpool', 4: 'upsample', 5: 'copy', 6: 'decode', 7: 'maxpool'}


class ConvDesc(C.Structure):
    _fields_ = [('x', C.c_void_p), ('w', C.c_void_p), ('bias', C.c_void_p), ('scale', C.c_void_p),
                ('shift', C.c_void_p), ('res', C.c_void_p), ('y', C.c_void_p),
                ('batch', C.c_int32), ('h', C.c_int32), ('w_in', C.c_int32), ('cin', C.c_int32), ('x_ld', C.c_int32),
                ('ho', C.c_int32), ('wo', C.c_int32), ('cout', C.c_int32), ('y_ld', C.c_int32), ('res_ld', C.c_int32),
                ('ksize', C.c_int32), ('stride', C.c_int32), ('pad', C.c_int32),
                ('kpad', C.c_int32), ('w_rows', C.c_int32), ('act', C.c_int32), ('slope', C.c_float),
                ('variant', C.c_int32), ('dtype', C.c_int32)]


class CspStage(C.Structure):
    _fields_ = [('w', C.c_void_p), ('bias', C.c_void_p), ('scale', C.c_void_p), ('shift', C.c_void_p),
                ('w_ld', C.c_int32), ('reserved', C.c_int32)]


class CspDesc(C.Structure):
    _fields_ = [('x', C.c_void_p), ('y', C.c_void_p), ('batch', C.c_int32), ('h', C.c_int32), ('w', C.c_int32),
                ('c1', C.c_int32), ('x_ld', C.c_int32), ('ch', C.c_int32), ('c2', C.c_int32), ('y_ld', C.c_int32),
                ('shortcut', C.c_int32), ('slope', C.c_float),
                ('cv1', CspStage), ('cv2', CspStage), ('m_cv1', CspStage), ('m_cv2', CspStage), ('cv3', CspStage), ('cv4', CspStage)]


class DecodeDesc(C.Structure):
    _fields_ = [('head', C.c_void_p), ('x_out', C.c_void_p), ('z_out', C.c_void_p),
                ('head_ld', C.c_int32),
                ('batch', C.c_int32), ('ny', C.c_int32), ('nx', C.c_int32), ('na', C.c_int32), ('no', C.c_int32),
                ('z_rows_total', C.c_int32), ('z_row0', C.c_int32),
                ('stride', C.c_float), ('anchor_wh', C.c_float * (2 * Y5_MAX_ANCHORS)), ('dtype', C.c_int32)]


class NmsDesc(C.Structure):
    _fields_ = [('pred', C.c_void_p), ('class_keep', C.c_void_p), ('workspace', C.c_void_p), ('det', C.c_void_p),
                ('count', C.c_void_p), ('kept_index', C.c_void_p), ('n_candidates', C.c_void_p),
                ('workspace_bytes', C.c_size_t), ('iou_thres', C.c_double), ('conf_thres', C.c_float),
                ('pred_dtype', C.c_int32), ('batch', C.c_int32), ('n_rows', C.c_int32), ('no', C.c_int32),
                ('agnostic', C.c_int32), ('max_candidates', C.c_int32), ('max_det', C.c_int32)]


# name -> (restype, argtypes); every symbol include/y5hip.h declares
_i, _p, _s = C.c_int, C.c_void_p, C.c_void_p
SIGNATURES = {
    'y5_version': (_i, []),
    'y5_last_error': (C.c_char_p, []),
    'y5_device_arch': (_i, [C.c_char_p, _i]),
    'y5_conv_f16': (_i, [C.POINTER(ConvDesc), _s]),
    'y5_focus_conv_f16': (_i, [C.POINTER(ConvDesc), _s]),
    'y5_focus_conv_u8': (_i, [C.POINTER(ConvDesc), _s]),
    'y5_focus_conv_u8_hwc': (_i, [C.POINTER(ConvDesc), _i, _s]),
    'y5_focus_conv2_f16': (_i, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), _s]),
    'y5_plan_add_focus_conv2': (_i, [_p, C.POINTER(ConvDesc), C.POINTER(ConvDesc)]),
    'y5_plan_add_focus_conv_u8_hwc': (_i, [_p, C.POINTER(ConvDesc), _i]),
    'y5_stem_conv_f16': (_i, [C.POINTER(ConvDesc), _s]),
    'y5_stem_conv_u8': (_i, [C.POINTER(ConvDesc), _s]),
    'y5_normalize_u8_f16': (_i, [C.c_void_p, C.c_void_p, C.c_int64, _s]),
    'y5_conv_num_variants': (_i, []),
    'y5_csp_f16': (_i, [C.POINTER(CspDesc), _s]),
    'y5_plan_add_csp': (_i, [_p, C.POINTER(CspDesc)]),
    'y5_conv_num_variants_bf16': (_i, []),
    'y5_conv_num_variants_phased': (_i, []),
    'y5_pack_image_f16': (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _s]),
    'y5_maxpool_f16': (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _s]),
    'y5_maxpool_cascade_f16': (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _s]),
    'y5_upsample2x_f16': (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _s]),
    'y5_maxpool_x16': (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _s]),
    'y5_maxpool_cascade_x16': (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _s]),
    'y5_upsample2x_x16': (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _s]),
    'y5_copy_channels_f16': (_i, [_p, _i, _p, _i, C.c_int64, _i, _s]),
    'y5_conv_stats_f16': (_i, [C.POINTER(ConvDesc), _p, _i, _i, _i, _s]),
    'y5_bn_stats_rows': (_i, [C.c_int64, _i]),
    'y5_bn_stats_f16': (_i, [_p, _i, C.c_int64, _i, _p, _s]),
    'y5_bn_finalize_f32': (_i, [_p, _i, C.c_int64, _i, _i, _p, _p, C.c_float, C.c_float, _p, _p, _p, _p, _s]),
    'y5_affine_act_f16': (_i, [_p, _i, _p, _i, _p, _i, C.c_int64, _i, _p, _p, _i, C.c_float, _s]),
    'y5_detect_decode_f16': (_i, [C.POINTER(DecodeDesc), _s]),
    'y5_detect_decode_multi_f16': (_i, [C.POINTER(DecodeDesc), C.c_int, _s]),
    'y5_detect_f16': (_i, [C.POINTER(ConvDesc), C.POINTER(DecodeDesc), C.c_int, _s]),
    'y5_detect_num_variants': (_i, []),
    'y5_conv_chain_f16': (_i, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), C.c_int, C.c_int, _s]),
    'y5_conv_chain_num_variants': (_i, []),
    'y5_conv_cat2_f16': (_i, [C.POINTER(ConvDesc), _p, _i, _i, _s]),
    'y5_conv_cat2_num_variants': (_i, []),
    'y5_sigmoid_f16': (_i, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, _s]),
    'y5_nms_workspace_bytes': (C.c_size_t, [_i, _i]),
    'y5_nms_batched': (_i, [C.POINTER(NmsDesc), _s]),
    'y5_nms_batched_noalloc': (_i, [C.POINTER(NmsDesc), _s]),
    'y5_match_predictions_f32': (_i, [C.c_void_p, C.c_void_p, C.c_void_p, _i, C.c_void_p, _i, C.c_void_p, C.c_void_p, _i, _i, _i,
                                      C.c_float, C.c_float, _s]),
    'y5_scale_coords_f32': (_i, [C.c_void_p, C.c_void_p, C.c_void_p, _i, _i, _i, _i, _s]),
    'y5_nms_boxes_f32': (_i, [_p, _p, _i, C.c_double, _p, _p, _i, _p, C.c_size_t, _s]),
    'y5_plan_create': (_i, [C.POINTER(_p)]),
    'y5_plan_destroy': (None, [_p]),
    'y5_plan_bind_slot': (_i, [_p, _i, _p, C.c_size_t]),
    'y5_plan_set_dtype': (_i, [_p, _i]),
    'y5_plan_add_conv': (_i, [_p, C.POINTER(ConvDesc)]),
    'y5_plan_add_focus_conv': (_i, [_p, C.POINTER(ConvDesc)]),
    'y5_plan_add_focus_conv_u8': (_i, [_p, C.POINTER(ConvDesc)]),
    'y5_plan_add_stem_conv': (_i, [_p, C.POINTER(ConvDesc)]),
    'y5_plan_add_stem_conv_u8': (_i, [_p, C.POINTER(ConvDesc)]),
    'y5_plan_add_normalize_u8': (_i, [_p, _p, _p, C.c_int64]),
    'y5_plan_add_pack_image': (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i]),
    'y5_plan_add_maxpool': (_i, [_p, _p, _i, _p, _i, _i, _i, _i, _i, _i]),
    'y5_plan_add_maxpool_cascade': (_i, [_p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i]),
    'y5_plan_add_upsample2x': (_i, [_p, _p, _i, _p, _i, _i, _i, _i, _i, _i]),
    'y5_plan_add_copy_channels': (_i, [_p, _p, _i, _p, _i, C.c_int64, _i]),
    'y5_plan_add_decode': (_i, [_p, C.POINTER(DecodeDesc)]),
    'y5_plan_add_decode_multi': (_i, [_p, C.POINTER(DecodeDesc), C.c_int]),
    'y5_plan_add_detect': (_i, [_p, C.POINTER(ConvDesc), C.POINTER(DecodeDesc), C.c_int]),
    'y5_plan_add_conv_chain': (_i, [_p, C.POINTER(ConvDesc), C.POINTER(ConvDesc), C.c_int, C.c_int]),
    'y5_plan_add_conv_cat2': (_i, [_p, C.POINTER(ConvDesc), _p, _i, _i]),
    'y5_plan_mark_last': (_i, [_p, C.c_int]),
    'y5_plan_num_ops': (_i, [_p]),
    'y5_plan_run': (_i, [_p, C.POINTER(_p), _i, _s]),
    'y5_plan_profile': (_i, [_p, C.POINTER(_p), _i, _s, C.POINTER(C.c_float), _i]),
    'y5_plan_op_info': (_i, [_p, _i, C.POINTER(_i), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

_lib = None


class Y5Error(RuntimeError):
    pass


def lib():
    """The loaded library; raises Y5Error with build instructions when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Y5Error('liby5hip.so not found at %s - build it with `make -C yolo_amd/csrc` '
                          '(hipcc --offload-arch=gfx950); the GPU path has no fallback' % LIB_PATH)
        # torch bundles its own HIP runtime (same soname as /opt/rocm's). Import torch first so that the dynamic
        # linker binds this library to the runtime instance torch uses; two runtimes in one process do not see
        # each other's devices, streams or allocations.
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here means the .so is stale
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        raise Y5Error('liby5hip error %d: %s' % (rc, lib().y5_last_error().decode('utf-8', 'replace')))


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
