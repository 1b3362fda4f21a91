"""Time every y5_conv_chain_f16 variant on a 3x3 (+ residual) -> 1x1 pair (hipEvents).
usage: bench_chain3x3.py [B] [C] [H] [W]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from yolo_amd import _lib
from test_gpu_conv import _desc

B, Cc, H, W = (list(map(int, sys.argv[1:])) + [32, 128, 40, 40][len(sys.argv) - 1:])[:4]
dev = torch.device('cuda:0')
lib = _lib.lib()
st = _lib.current_stream()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
x = torch.randn(B, H, W, Cc, device=dev).half()
res = torch.randn(B, H, W, Cc, device=dev).half()
w1 = (torch.randn(Cc, Cc, 3, 3, device=dev) / (9 * Cc) ** 0.5).half()
w2 = (torch.randn(Cc, Cc, 1, 1, device=dev) / Cc ** 0.5).half()
b1, b2 = torch.randn(Cc, device=dev), torch.randn(Cc, device=dev)
d1, y1, k1 = _desc(dev, x, w1, bias=b1, slope=0.1, res=res)
d2, y2, k2 = _desc(dev, y1, w2, bias=b2, slope=0.1)
flops = 2.0 * B * H * W * Cc * Cc * 10
for keep in (1, 0):
    for v in range(1, lib.y5_conv_chain_num_variants() + 1):
        if lib.y5_conv_chain_f16(C.byref(d1), C.byref(d2), keep, v, st) != 0:
            continue
        lib.y5_conv_chain_f16(C.byref(d1), C.byref(d2), keep, v, st)
        e0.record()
        for _ in range(20):
            lib.y5_conv_chain_f16(C.byref(d1), C.byref(d2), keep, v, st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print('keep_first %d  v%-2d %.1f us  %.0f TFLOP/s' % (keep, v, us, flops / us / 1e6))
