"""Time one convolution shape through y5_conv_f16 (hipEvents), for kernel tuning and rocprofv3 counter runs.
usage: bench_conv.py B Cin Cout H W k stride [iters] [variant | all]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yolo_amd import _lib

B, Ci, Co, H, W, k, s = map(int, sys.argv[1:8])
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 20
dev = torch.device('cuda:0')
lib = _lib.lib()
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
d.act, d.slope = _lib.ACT_LEAKY, 0.1
st = _lib.current_stream()
flops = 2.0 * B * Ho * Wo * Co * k * k * Ci
nbytes = 2.0 * (B * H * W * Ci + B * Ho * Wo * Co)
vsel = sys.argv[9] if len(sys.argv) > 9 else '0'
if vsel == 'all':
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for v in range(lib.y5_conv_num_variants() + 1):
        d.variant = v
        if lib.y5_conv_f16(C.byref(d), st) != 0:
            continue
        lib.y5_conv_f16(C.byref(d), st)
        e0.record()
        for _ in range(iters):
            lib.y5_conv_f16(C.byref(d), st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        print('  v%-2d %.1f us  %.1f TFLOP/s  %.0f GB/s(in+out)' % (v, us, flops / us / 1e6, nbytes / us / 1e3))
    sys.exit(0)
d.variant = int(vsel)
for _ in range(3):
    _lib.check(lib.y5_conv_f16(C.byref(d), st))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    _lib.check(lib.y5_conv_f16(C.byref(d), st))
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / iters * 1e3
print('conv B%d %d->%d %dx%d k%d s%d: %.1f us  %.1f TFLOP/s  %.0f GB/s(in+out)' % (B, Ci, Co, H, W, k, s, us, flops / us / 1e6, nbytes / us / 1e3))
