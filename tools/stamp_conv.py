"""Phase breakdown of the im2col convolution kernel's K loop from in-kernel cycle stamps (diagnostic library built by
`make -C yolo_amd/csrc dbg`).  usage: stamp_conv.py B Cin Cout H W k stride variant"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from yolo_amd import _lib

lib = C.CDLL(os.path.join(ROOT, 'yolo_amd', 'liby5hip_dbg.so'))
B, Ci, Co, H, W, k, s, var = map(int, sys.argv[1:9])
dev = torch.device('cuda:0')
cip, cop = (Ci + 7) // 8 * 8, (Co + 7) // 8 * 8
Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
x = torch.randn(B, H, W, cip, device=dev).half()
kk = k * k * cip
kpad, rows = (kk + 63) // 64 * 64, (Co + 255) // 256 * 256
w = torch.zeros(rows, kpad, dtype=torch.float16, device=dev)
w[:Co, :kk] = (torch.randn(Co, kk, device=dev) / kk ** 0.5).half()
y = torch.zeros(B, Ho, Wo, cop, dtype=torch.float16, device=dev)
bias = torch.randn(cop, device=dev)
d = _lib.ConvDesc()
d.x, d.w, d.y, d.bias = x.data_ptr(), w.data_ptr(), y.data_ptr(), bias.data_ptr()
d.batch, d.h, d.w_in, d.cin, d.x_ld = B, H, W, cip, cip
d.ho, d.wo, d.cout, d.y_ld = Ho, Wo, cop, cop
d.ksize, d.stride, d.pad, d.kpad, d.w_rows = k, s, k // 2, kpad, rows
d.act, d.slope, d.variant = 1, 0.1, var
if os.environ.get('STAGING_ONLY'):
    d.res_ld = -12345
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
print('%.1f us per launch%s' % (e0.elapsed_time(e1) / 20 * 1e3, ' (staging only)' if os.environ.get('STAGING_ONLY') else ''))
n = 65536
force = os.environ.get('STAMP_KIND')  # 'igemm' reads the im2col kernel's stamps whatever the variant id
if hasattr(lib, 'y5_debug_stamps3') and var >= 45 and not force:
    buf = np.zeros((n, 8), dtype=np.uint64)
    assert lib.y5_debug_stamps3(buf.ctypes.data_as(C.c_void_p), n) == 0
    v = buf[buf[:, 6] > 0].astype(np.float64)
    nt = v[:, 6].mean()
    print('variant %d: %d waves, %.2f tiles per workgroup' % (var, len(v), nt))
    for name, col in (('products', 0), ('stage+fetch', 1), ('epilogue', 2), ('barrier', 3)):
        print('  %-14s mean %9.0f  per tile %8.1f' % (name, v[:, col].mean(), (v[:, col] / v[:, 6]).mean()))
    print('  prologue       mean %9.0f;  whole kernel %9.0f' % (v[:, 5].mean(), v[:, 4].mean()))
    sys.exit(0)
buf = np.zeros((n, 8), dtype=np.uint64)
if 39 <= var <= 41 and not force:
    assert lib.y5_debug_stampsh(buf.ctypes.data_as(C.c_void_p), n) == 0
else:
    assert lib.y5_debug_stamps(buf.ctypes.data_as(C.c_void_p), n) == 0
v = buf[buf[:, 6] > 0].astype(np.float64)
nk = v[0, 6]
t0 = v[:, 5].min()
span = (v[:, 5] + v[:, 3] + v[:, 4]).max() - t0
print('variant %d: %d waves stamped, nk %d; kernel span %.0f cycles (100 MHz ticks?)' % (var, len(v), nk, span))
for name, col in (('wait+barrier', 0), ('dma issue', 1), ('lds+mfma', 2), ('k loop total', 3), ('epilogue', 4), ('2nd barrier', 7)):
    print('  %-14s mean %9.0f  per step %8.1f   (p10 %.0f p90 %.0f)' % (name, v[:, col].mean(), v[:, col].mean() / nk,
                                                                    np.percentile(v[:, col], 10), np.percentile(v[:, col], 90)))
print('  wave start spread: p50 %.0f p90 %.0f max %.0f' % tuple(np.percentile(v[:, 5] - t0, [50, 90, 100])))
