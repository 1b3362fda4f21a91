"""Time the SPP pooling cascade (y5_maxpool_cascade_f16) on the concat buffer of yolov5s layer 8 (hipEvents).
usage: bench_pool.py [B] [C] [H] [W] [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yolo_amd import _lib

B, c, h, w, iters = (list(map(int, sys.argv[1:])) + [32, 256, 20, 20, 50][len(sys.argv) - 1:])[:5]
dev = torch.device('cuda:0')
lib = _lib.lib()
cat = torch.randn(B, h, w, 4 * c, device=dev).half()
st = _lib.current_stream()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    e0.record()
    for _ in range(iters):
        _lib.check(lib.y5_maxpool_cascade_f16(cat.data_ptr(), 4 * c, cat.data_ptr() + 2 * c, 4 * c, c, 3, B, h, w, c, 5, st))
    e1.record()
    torch.cuda.synchronize()
us = e0.elapsed_time(e1) / iters * 1e3
print('pool cascade B%d c%d %dx%d: %.1f us  %.0f GB/s' % (B, c, h, w, us, B * h * w * c * 2 * 4 / us / 1e3))
