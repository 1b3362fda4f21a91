"""Where a linear-halo 3x3 workgroup spends its time, from in-kernel cycle stamps (diagnostic library built by
`make -C yolo_amd/csrc dbg`).  usage: stamp_lin.py B C Cout H W variant"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from yolo_amd import _lib

lib = C.CDLL(os.path.join(ROOT, 'yolo_amd', 'liby5hip_dbg.so'))
B, Ci, Co, H, W, var = map(int, sys.argv[1:7])
dev = torch.device('cuda:0')
x = torch.randn(B, H, W, Ci, device=dev).half()
kk = 9 * Ci
kpad, rows = (kk + 63) // 64 * 64, (Co + 255) // 256 * 256
w = torch.zeros(rows, kpad, dtype=torch.float16, device=dev)
w[:Co, :kk] = (torch.randn(Co, kk, device=dev) / kk ** 0.5).half()
y = torch.zeros(B, H, W, Co, dtype=torch.float16, device=dev)
bias = torch.randn(Co, device=dev)
d = _lib.ConvDesc()
d.x, d.w, d.y, d.bias = x.data_ptr(), w.data_ptr(), y.data_ptr(), bias.data_ptr()
d.batch, d.h, d.w_in, d.cin, d.x_ld = B, H, W, Ci, Ci
d.ho, d.wo, d.cout, d.y_ld = H, W, Co, Co
d.ksize, d.stride, d.pad, d.kpad, d.w_rows = 3, 1, 1, kpad, rows
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
print('%.1f us per launch' % (e0.elapsed_time(e1) / 20 * 1e3))
n = 65536
buf = np.zeros((n, 8), dtype=np.uint64)
assert lib.y5_debug_stamps_lin(buf.ctypes.data_as(C.c_void_p), n) == 0
v = buf[buf[:, 5] > 0].astype(np.float64)
t0 = v[:, 0].min()
nk = v[0, 5]
print('%d consumer waves, nk %d; kernel span %.0f ticks (first start -> last end)' % (len(v), nk, v[:, 3].max() - t0))
for name, col in (('start', 0), ('halo+stage0 landed', 1), ('K loop end', 2), ('end', 3)):
    c = v[:, col] - t0
    print('  %-20s p10 %8.0f  p50 %8.0f  p90 %8.0f  max %8.0f' % (name, *np.percentile(c, [10, 50, 90, 100])))
print('  prologue (start -> first barrier passed): mean %.0f' % (v[:, 1] - v[:, 0]).mean())
print('  K loop: mean %.0f = %.1f per step; of it in barriers %.0f = %.1f per step' % ((v[:, 2] - v[:, 1]).mean(), (v[:, 2] - v[:, 1]).mean() / nk, v[:, 4].mean(), v[:, 4].mean() / nk))
print('  epilogue: mean %.0f' % (v[:, 3] - v[:, 2]).mean())
