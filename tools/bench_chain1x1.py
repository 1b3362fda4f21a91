"""Time the persistent 1x1 chain (y5_conv_chain_f16, last variant but one) against its two convolutions as separate
y5_conv_f16 launches, each with its fastest variant (hipEvents).  usage: bench_chain1x1.py [B]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from yolo_amd import _lib
from test_gpu_conv import _desc

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda:0')
lib = _lib.lib()
st = _lib.current_stream()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn, iters=30):
    fn(); fn()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def best_variant(d):
    best = (1e9, 0)
    for v in range(lib.y5_conv_num_variants() + 1):
        d.variant = v
        if lib.y5_conv_f16(C.byref(d), st) != 0:
            continue
        best = min(best, (timed(lambda: lib.y5_conv_f16(C.byref(d), st), 10), v))
    d.variant = best[1]
    return best


for Ci, Cm, Cs, Co, H in ((256, 256, 128, 128, 40), (128, 128, 64, 64, 80)):
    x = torch.randn(B, H, H, Ci, device=dev).half()
    w1 = (torch.randn(Cm, Ci, 1, 1, device=dev) / Ci ** 0.5).half()
    w2 = (torch.randn(Co, Cs, 1, 1, device=dev) / Cs ** 0.5).half()
    b1, b2 = torch.randn(Cm, device=dev), torch.randn(Co, device=dev)
    d1, y1, k1 = _desc(dev, x, w1, bias=b1, scale=b1.abs() + 0.5, shift=b1, slope=0.1)
    d2, y2, k2 = _desc(dev, y1[..., :Cs], w2, bias=b2, slope=0.1)
    d2.x_ld = Cm
    t1, v1 = best_variant(d1)
    t2, v2 = best_variant(d2)
    v = lib.y5_conv_chain_num_variants() - 1
    tc = timed(lambda: _lib.check(lib.y5_conv_chain_f16(C.byref(d1), C.byref(d2), 1, v, st)))
    tc0 = timed(lambda: _lib.check(lib.y5_conv_chain_f16(C.byref(d1), C.byref(d2), 0, v, st)))
    print('%d->%d then %d->%d @%dx%d B%d: separate %.1f (v%d) + %.1f (v%d) = %.1f us; chain %.1f us (first result not stored: %.1f us)'
          % (Ci, Cm, Cs, Co, H, H, B, t1, v1, t2, v2, t1 + t2, tc, tc0))
