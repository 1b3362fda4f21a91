"""Time one Detect level through y5_detect_f16 (every tile configuration, with and without the raw copy x_i) and
through the two-kernel path.  usage: bench_detect.py B Cin ny nx"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yolo_amd import _lib

B, Ci, ny, nx = map(int, sys.argv[1:5])
na, no, cpad = 3, 85, 256
dev = torch.device('cuda:0')
lib = _lib.lib()
x = torch.randn(B, ny, nx, Ci, device=dev).half()
w = torch.zeros(256, (Ci + 63) // 64 * 64, dtype=torch.float16, device=dev)
w[:255, :Ci] = (torch.randn(255, Ci, device=dev) / Ci ** 0.5).half()
bias = torch.randn(cpad, device=dev)
rows = na * ny * nx
xo = torch.zeros(B, na, ny, nx, no, dtype=torch.float16, device=dev)
z = torch.zeros(B, rows, no, dtype=torch.float16, device=dev)
head = torch.zeros(B, ny, nx, cpad, dtype=torch.float16, device=dev)
st = _lib.current_stream()
cd = _lib.ConvDesc()
cd.x, cd.w, cd.bias, cd.y = x.data_ptr(), w.data_ptr(), bias.data_ptr(), head.data_ptr()
cd.batch, cd.h, cd.w_in, cd.cin, cd.x_ld = B, ny, nx, Ci, Ci
cd.ho, cd.wo, cd.cout, cd.y_ld = ny, nx, cpad, cpad
cd.ksize, cd.stride, cd.pad, cd.kpad, cd.w_rows = 1, 1, 0, w.shape[1], 256
cd.act, cd.slope = _lib.ACT_NONE, 0.0


def dec(with_x, with_head):
    d = _lib.DecodeDesc()
    d.z_out = z.data_ptr()
    if with_x:
        d.x_out = xo.data_ptr()
    if with_head:
        d.head, d.head_ld = head.data_ptr(), cpad
    d.batch, d.ny, d.nx, d.na, d.no = B, ny, nx, na, no
    d.z_rows_total, d.z_row0, d.stride = rows, 0, 8.0
    for a in range(6):
        d.anchor_wh[a] = 10.0 + a
    return d


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


mb = (x.numel() + 2 * z.numel()) * 2 / 1e6
d2 = dec(True, True)
def two():
    _lib.check(lib.y5_conv_f16(C.byref(cd), st)); _lib.check(lib.y5_detect_decode_f16(C.byref(d2), st))
print('conv + decode (two kernels): %.1f us' % timed(two))
for with_x in (True, False):
    d1 = dec(with_x, False)
    for v in range(1, lib.y5_detect_num_variants() + 1):
        us = timed(lambda: _lib.check(lib.y5_detect_f16(C.byref(cd), C.byref(d1), v, st)))
        print('one kernel v%d %s: %.1f us  %.0f GB/s' % (v, 'z + x' if with_x else 'z only', us, (mb if with_x else mb - z.numel() * 2 / 1e6) / us * 1e3))
