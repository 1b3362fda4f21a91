"""The C-ABI library loads without a GPU, exports every symbol include/y5hip.h declares, and reports argument
errors through return codes + y5_last_error() (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

from util import ROOT
from yolo_amd import _lib


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'y5hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(y5_[a-z0-9_]+)\s*\(', text)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), 'liby5hip.so does not export %s' % name
        assert name in _lib.SIGNATURES, 'yolo_amd/_lib.py does not bind %s' % name
    assert sorted(_lib.SIGNATURES) == names


def test_version_and_error_channel():
    lib = _lib.lib()
    assert lib.y5_version() >= 100
    rc = lib.y5_conv_f16(None, None)
    assert rc == -1 and b'null descriptor' in lib.y5_last_error()
    d = _lib.ConvDesc()
    rc = lib.y5_conv_f16(C.byref(d), None)
    assert rc == -1 and b'null tensor pointer' in lib.y5_last_error()
    with pytest.raises(_lib.Y5Error):
        _lib.check(rc)


def test_descriptor_validation_happens_before_any_launch():
    lib = _lib.lib()
    d = _lib.ConvDesc()
    d.x = d.w = d.y = 0x1000
    d.ksize, d.stride, d.pad = 5, 1, 2
    assert lib.y5_conv_f16(C.byref(d), None) == -1 and b'ksize' in lib.y5_last_error()
    dd = _lib.DecodeDesc()
    dd.head = dd.z_out = 0x1000
    dd.batch = dd.ny = dd.nx = 1
    dd.na, dd.no = 99, 85
    assert lib.y5_detect_decode_f16(C.byref(dd), None) == -1 and b'na 99' in lib.y5_last_error()
    n = _lib.NmsDesc()
    assert lib.y5_nms_batched(C.byref(n), None) == -1
    assert lib.y5_nms_workspace_bytes(2, 1000) == 256 + 2 * 1024 * 8
    assert lib.y5_maxpool_f16(0x1000, 8, 0x2000, 8, 1, 4, 4, 8, 4, None) == -1 and b'odd' in lib.y5_last_error()


def test_plan_objects_are_host_only():
    lib = _lib.lib()
    h = C.c_void_p()
    _lib.check(lib.y5_plan_create(C.byref(h)))
    assert lib.y5_plan_num_ops(h) == 0
    _lib.check(lib.y5_plan_bind_slot(h, 0, 0x10000, 4096))
    assert lib.y5_plan_bind_slot(h, 99, 0x10000, 4096) == -1
    _lib.check(lib.y5_plan_add_copy_channels(h, 0x10000, 8, 0x20000, 8, 16, 8))
    assert lib.y5_plan_num_ops(h) == 1
    kind, flops, nbytes = C.c_int(), C.c_double(), C.c_double()
    _lib.check(lib.y5_plan_op_info(h, 0, C.byref(kind), C.byref(flops), C.byref(nbytes)))
    assert kind.value == 5 and nbytes.value == 4.0 * 16 * 8
    # the op reads external slot 0, so running without it must be refused before anything is launched
    assert lib.y5_plan_run(h, None, 0, None) == -1 and b'external slot' in lib.y5_last_error()
    lib.y5_plan_destroy(h)
