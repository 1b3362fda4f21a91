"""Where the weights-in-registers 128-channel 3x3 + 1x1 kernel spends its time, from in-kernel cycle stamps (diagnostic
library built by `make -C yolo_amd/csrc dbg`).  usage: stamp_c128.py [batch] [H] [W]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from yolo_amd import _lib

lib = C.CDLL(os.path.join(ROOT, 'yolo_amd', 'liby5hip_dbg.so'))
B, H, W = (list(map(int, sys.argv[1:])) + [32, 40, 40][len(sys.argv) - 1:])[:3]
dev = torch.device('cuda:0')
c = 128
x = torch.randn(B, H, W, c, device=dev).half()
res = torch.randn(B, H, W, c, device=dev).half()
y1 = torch.zeros(B, H, W, c, dtype=torch.float16, device=dev)
y2 = torch.zeros(B, H, W, c, dtype=torch.float16, device=dev)
w1 = torch.zeros(256, 9 * c, dtype=torch.float16, device=dev); w1[:c] = (torch.randn(c, 9 * c, device=dev) / 34).half()
w2 = torch.zeros(256, c, dtype=torch.float16, device=dev); w2[:c] = (torch.randn(c, c, device=dev) / 11).half()
b1, b2 = torch.randn(c, device=dev) * 0.1, torch.randn(c, device=dev) * 0.1
d1, d2 = _lib.ConvDesc(), _lib.ConvDesc()
d1.x, d1.w, d1.y, d1.bias, d1.res, d1.res_ld = x.data_ptr(), w1.data_ptr(), y1.data_ptr(), b1.data_ptr(), res.data_ptr(), c
d1.batch, d1.h, d1.w_in, d1.cin, d1.x_ld, d1.ho, d1.wo, d1.cout, d1.y_ld = B, H, W, c, c, H, W, c, c
d1.ksize, d1.stride, d1.pad, d1.kpad, d1.w_rows, d1.act, d1.slope = 3, 1, 1, 9 * c, 256, 1, 0.1
d2.x, d2.w, d2.y, d2.bias = y1.data_ptr(), w2.data_ptr(), y2.data_ptr(), b2.data_ptr()
d2.batch, d2.h, d2.w_in, d2.cin, d2.x_ld, d2.ho, d2.wo, d2.cout, d2.y_ld = B, H, W, c, c, H, W, c, c
d2.ksize, d2.stride, d2.pad, d2.kpad, d2.w_rows, d2.act, d2.slope = 1, 1, 0, c, 256, 1, 0.1
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
lib.y5_conv_chain_f16.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
lib.y5_last_error.restype = C.c_char_p
v = lib.y5_conv_chain_num_variants() - 2
for _ in range(3):
    assert lib.y5_conv_chain_f16(C.byref(d1), C.byref(d2), 1, v, st) == 0, lib.y5_last_error()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    lib.y5_conv_chain_f16(C.byref(d1), C.byref(d2), 1, v, st)
e1.record(); torch.cuda.synchronize()
print('%.1f us per launch' % (e0.elapsed_time(e1) / 20 * 1e3))
buf = np.zeros((256 * 8, 12), dtype=np.uint64)
assert lib.y5_debug_stamps_c128(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
s = buf[buf[:, 11] > 0].astype(np.float64)
names = ['weights + parameters asked', 'copies issued', 'everything landed (vmcnt 0)', 'first barrier', 'products (all passes)', 'barrier 1',
         'sum + post-ops', 'barrier 2', '1x1 + stores']
print('%d waves; ticks per workgroup, mean over waves (kh = 0 waves | kh = 1 waves):' % len(s))
k0 = s[np.arange(len(s)) % 8 < 4]
k1 = s[np.arange(len(s)) % 8 >= 4]
for i, n in enumerate(names):
    print('  %-28s %9.1f   (%9.1f | %9.1f)' % (n, s[:, i].mean(), k0[:, i].mean(), k1[:, i].mean()))
print('  total                        %9.1f' % s[:, :9].sum(1).mean())
