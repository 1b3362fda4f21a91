"""Where the one-kernel Focus + stride-2 conv spends its time, from in-kernel cycle stamps (diagnostic library built by
`make -C yolo_amd/csrc dbg`).  usage: stamp_stem2.py [batch] [H] [W] [variant]
With the production library: launch time of the chosen variant, and its output compared bit for bit with variant 0's."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from yolo_amd import _lib

lib = C.CDLL(os.path.join(ROOT, 'yolo_amd', os.environ.get('Y5_DBG_LIB', 'liby5hip_dbg.so')))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H = int(sys.argv[2]) if len(sys.argv) > 2 else 640
W = int(sys.argv[3]) if len(sys.argv) > 3 else 640
VARIANT = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = torch.device('cuda:0')
img = torch.rand(B, 3, H, W, device=dev).half()
w0 = torch.zeros(128, 192, dtype=torch.float16, device=dev); w0[:32, :144] = (torch.randn(32, 144, device=dev) / 10).half()
w1 = torch.zeros(256, 320, dtype=torch.float16, device=dev); w1[:64, :288] = (torch.randn(64, 288, device=dev) / 17).half()
b0, b1 = torch.randn(32, device=dev) * 0.1, torch.randn(64, device=dev) * 0.1
y = torch.zeros(B, H // 4, W // 4, 64, dtype=torch.float16, device=dev)
d0, d1 = _lib.ConvDesc(), _lib.ConvDesc()
d0.x, d0.w, d0.bias = img.data_ptr(), w0.data_ptr(), b0.data_ptr()
d0.batch, d0.h, d0.w_in, d0.cin, d0.ho, d0.wo, d0.cout, d0.y_ld = B, H, W, 3, H // 2, W // 2, 32, 32
d0.ksize, d0.stride, d0.pad, d0.kpad, d0.w_rows, d0.act, d0.slope = 3, 1, 1, 192, 128, 1, 0.1
d1.w, d1.y, d1.bias = w1.data_ptr(), y.data_ptr(), b1.data_ptr()
d1.batch, d1.h, d1.w_in, d1.cin, d1.x_ld, d1.ho, d1.wo, d1.cout, d1.y_ld = B, H // 2, W // 2, 32, 32, H // 4, W // 4, 64, 64
d1.ksize, d1.stride, d1.pad, d1.kpad, d1.w_rows, d1.act, d1.slope = 3, 2, 1, 320, 256, 1, 0.1
d0.variant = VARIANT
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
lib.y5_focus_conv2_f16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
lib.y5_last_error.restype = C.c_char_p
for _ in range(3):
    assert lib.y5_focus_conv2_f16(C.byref(d0), C.byref(d1), st) == 0, lib.y5_last_error()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    lib.y5_focus_conv2_f16(C.byref(d0), C.byref(d1), st)
e1.record(); torch.cuda.synchronize()
print('%.1f us per launch' % (e0.elapsed_time(e1) / 20 * 1e3))
if not hasattr(lib, 'y5_debug_stamps_stem2'):  # the production library: launch time, and the variants agree
    got = y.clone()
    y.zero_()
    d0.variant = 0
    assert lib.y5_focus_conv2_f16(C.byref(d0), C.byref(d1), st) == 0, lib.y5_last_error()
    torch.cuda.synchronize()
    print('variant %d == variant 0: %s (|y| mean %.3f)' % (VARIANT, bool(torch.equal(got, y)), float(y.float().abs().mean())))
    sys.exit(0)
buf = np.zeros((256 * 8, 8), dtype=np.uint64)
assert lib.y5_debug_stamps_stem2(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
s = buf[buf[:, 7] > 0].astype(np.float64)
print('%d waves, %.2f tiles each; ticks per tile, mean over waves:' % (len(s), s[:, 7].mean()))
for i, n in enumerate(['barrier A', 'layer 0', 'barrier B', 'layer 1 + stores', 'copy issue + locate']):
    print('  %-18s %8.1f   (waves 0-2: %8.1f, waves 3-7: %8.1f)' % (n, (s[:, i] / s[:, 7]).mean(), (s[0::8, i] / s[0::8, 7]).mean(), (s[7::8, i] / s[7::8, 7]).mean()))
print('  total              %8.1f' % (s[:, :5].sum(1) / s[:, 7]).mean())
