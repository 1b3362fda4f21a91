"""Where the one-kernel BottleneckCSP spends its time, from in-kernel cycle stamps (diagnostic library built by
`make -C yolo_amd/csrc dbg`).  usage: stamp_csp.py [batch]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from yolo_amd import _lib

lib = C.CDLL(os.path.join(ROOT, 'yolo_amd', 'liby5hip_dbg.so'))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
H = W = 160
dev = torch.device('cuda:0')
x = torch.randn(B, H, W, 64, device=dev).half()
y = torch.zeros(B, H, W, 64, dtype=torch.float16, device=dev)
keep = []
def stage(rows, k, bias=True, scale=False):
    kpad = (k + 63) // 64 * 64
    w = torch.zeros(256, kpad, dtype=torch.float16, device=dev)
    w[:rows, :k] = (torch.randn(rows, k, device=dev) / k ** 0.5).half()
    b = torch.randn(rows, device=dev) * 0.1
    keep.extend([w, b])
    st = _lib.CspStage()
    st.w, st.w_ld = w.data_ptr(), kpad
    if bias:
        st.bias = b.data_ptr()
    if scale:
        sc, sh = torch.rand(rows, device=dev) + 0.5, torch.randn(rows, device=dev) * 0.1
        keep.extend([sc, sh])
        st.scale, st.shift = sc.data_ptr(), sh.data_ptr()
    return st
d = _lib.CspDesc()
d.x, d.y, d.batch, d.h, d.w = x.data_ptr(), y.data_ptr(), B, H, W
d.c1, d.x_ld, d.ch, d.c2, d.y_ld, d.shortcut, d.slope = 64, 64, 32, 64, 64, 1 | int(os.environ.get('CSP_DIAG', '0')), 0.1
d.cv1, d.cv2, d.m_cv1, d.m_cv2, d.cv3, d.cv4 = stage(32, 64), stage(32, 64, False, True), stage(32, 32), stage(32, 288), stage(32, 32, False, True), stage(64, 64)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
lib.y5_csp_f16.argtypes = [C.c_void_p, C.c_void_p]
lib.y5_last_error.restype = C.c_char_p
for _ in range(3):
    rc = lib.y5_csp_f16(C.byref(d), st)
    assert rc == 0, lib.y5_last_error()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    lib.y5_csp_f16(C.byref(d), st)
e1.record()
torch.cuda.synchronize()
print('%.1f us per launch' % (e0.elapsed_time(e1) / 20 * 1e3))
nw = 8
buf = np.zeros((256 * nw, 8), dtype=np.uint64)
assert lib.y5_debug_stamps_csp(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
v = buf[buf[:, 7] > 0].astype(np.float64)
names = ['top wait', 'copy issue', 'stage 1 [t|y2]', 'barrier', 'stage 2 u', 'barrier', 'rows (3x3, cv3, cv4)']
print('%d waves, %.1f tiles each; ticks (100 MHz) per tile, mean over waves:' % (len(v), v[:, 7].mean()))
for i, n in enumerate(names):
    print('  %-22s %8.1f' % (n, (v[:, i] / v[:, 7]).mean()))
print('  total                  %8.1f' % (v[:, :7].sum(1) / v[:, 7]).mean())
