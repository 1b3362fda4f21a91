"""Time every y5_conv_cat2_f16 variant (1x1 convolution of [nearest-2x-upsampled a | b]) on one shape (hipEvents).
usage: bench_cat2.py [B] [Ca] [Cb] [Co] [H] [W]   (H, W: the output = b's size)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from yolo_amd import _lib
from test_gpu_conv import _desc

B, Ca, Cb, Co, H, W = (list(map(int, sys.argv[1:])) + [32, 128, 128, 128, 80, 80][len(sys.argv) - 1:])[:6]
dev = torch.device('cuda:0')
lib = _lib.lib()
st = _lib.current_stream()
a = torch.randn(B, H // 2, W // 2, Ca, device=dev).half()
b = torch.randn(B, H, W, Cb, device=dev).half()
w = (torch.randn(Co, Ca + Cb, 1, 1, device=dev) / (Ca + Cb) ** 0.5).half()
bias = torch.randn(Co, device=dev)
shape_only = torch.empty(B, H, W, Ca + Cb, dtype=torch.float16, device='meta')
d, y, keep = _desc(dev, torch.zeros(B, H, W, Ca + Cb, dtype=torch.float16, device=dev), w, bias=bias, slope=0.1)
d.x, d.cin, d.x_ld = a.data_ptr(), Ca, Ca
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
mb = (a.numel() + b.numel() + y.numel()) * 2 / 1e6
for v in range(1, lib.y5_conv_cat2_num_variants() + 1):
    d.variant = v
    if lib.y5_conv_cat2_f16(C.byref(d), b.data_ptr(), Cb, Cb, st) != 0:
        continue
    lib.y5_conv_cat2_f16(C.byref(d), b.data_ptr(), Cb, Cb, st)
    e0.record()
    for _ in range(20):
        lib.y5_conv_cat2_f16(C.byref(d), b.data_ptr(), Cb, Cb, st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print('v%-2d %.1f us  %.0f GB/s' % (v, us, mb / us * 1e3))
