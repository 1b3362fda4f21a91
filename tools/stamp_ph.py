"""Where a workgroup of the phased convolution kernel (conv_ph.hip) spends its time per tile, from in-kernel cycle stamps
(diagnostic library: `make -C yolo_amd/csrc dbg`).  usage: stamp_ph.py B Cin Cout H W k stride variant"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from yolo_amd import _lib

lib = C.CDLL(os.path.join(ROOT, 'yolo_amd', 'liby5hip_dbg.so'))
B, Ci, Co, H, W, k, s, var = map(int, sys.argv[1:9])
dev = torch.device('cuda:0')
Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
x = torch.randn(B, H, W, Ci, device=dev).half()
kk = k * k * Ci
kpad, rows = (kk + 63) // 64 * 64, (Co + 255) // 256 * 256
w = torch.zeros(rows, kpad, dtype=torch.float16, device=dev)
w[:Co, :kk] = (torch.randn(Co, kk, device=dev) / kk ** 0.5).half()
y = torch.zeros(B, Ho, Wo, Co, dtype=torch.float16, device=dev)
bias = torch.randn(Co, device=dev)
d = _lib.ConvDesc()
d.x, d.w, d.y, d.bias = x.data_ptr(), w.data_ptr(), y.data_ptr(), bias.data_ptr()
d.batch, d.h, d.w_in, d.cin, d.x_ld = B, H, W, Ci, Ci
d.ho, d.wo, d.cout, d.y_ld = Ho, Wo, Co, Co
d.ksize, d.stride, d.pad, d.kpad, d.w_rows = k, s, k // 2, kpad, rows
d.act, d.slope, d.variant = 1, 0.1, var
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
lib.y5_conv_f16.argtypes = [C.c_void_p, C.c_void_p]
for _ in range(5):
    assert lib.y5_conv_f16(C.byref(d), st) == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    lib.y5_conv_f16(C.byref(d), st)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print('%.1f us per launch, %.1f TFLOP/s' % (us, 2.0 * B * Ho * Wo * Co * kk / us / 1e6))
n = 65536
buf = np.zeros((n, 8), dtype=np.uint64)
assert lib.y5_debug_stamps_ph(buf.ctypes.data_as(C.c_void_p), n) == 0
v = buf[buf[:, 5] > 0].astype(np.float64)
t0 = v[:, 0].min()
clk = (v[:, 5] - v[:, 0]) / (v[:, 7] - v[:, 6]) * 100.0
print('%d tiles stamped; launch span %.0f ticks; in-kernel clock %.0f MHz (median)' % (len(v), v[:, 5].max() - t0, np.median(clk)))
names = ['set-up', 'prologue (issue -> K-tile 0 landed)', 'K loop', 'post-ops -> LDS tile', 'stores issued']
for i, name in enumerate(names):
    c = v[:, i + 1] - v[:, i]
    print('  %-38s mean %8.0f  p10 %8.0f  p90 %8.0f ticks' % (name, c.mean(), *np.percentile(c, [10, 90])))
tot = (v[:, 5] - v[:, 0])
print('  %-38s mean %8.0f ticks = %.1f us at the median clock' % ('tile total', tot.mean(), tot.mean() / np.median(clk)))
sec = np.zeros((n, 2, 8), dtype=np.uint64)
assert lib.y5_debug_sections_ph(sec.ctypes.data_as(C.c_void_p), n) == 0
sec = sec[buf[:, 5] > 0].astype(np.float64)
nk = (kk + 63) // 64
for g, name in ((0, 'wave 0'), (1, 'wave 4')):
    m = sec[:, g, :5].mean(axis=0) / nk
    print('  %s per K-tile: issue + reads %.0f, vmcnt wait %.0f, lgkmcnt + barrier %.0f, MFMAs %.0f, second barrier %.0f = %.0f ticks'
          % (name, m[0], m[1], m[2], m[3], m[4], m.sum()))
