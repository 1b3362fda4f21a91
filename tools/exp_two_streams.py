"""Experiment: one batch-32 forward against two batch-16 forwards issued on two streams."""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import util
from yolo_amd import engine

torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
m = util.build_model('yolov5s'); util.calibrate(m, size=640); m = util.quiet(m.fuse); m = m.eval().half().to(dev)
B = 32
x = torch.rand(B, 3, 640, 640).half().to(dev)
for _ in range(3): m(x)
def timeit(fn, n=20):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print('one stream, batch 32: %.3f ms' % timeit(lambda: m(x)))
for parts in (2, 4):
    ms = [copy.deepcopy(m) for _ in range(parts)]
    xs = list(x.chunk(parts))
    streams = [torch.cuda.Stream() for _ in range(parts)]
    for mm, xx in zip(ms, xs):
        for _ in range(3): mm(xx)
    print('one stream, batch %d: %.3f ms' % (B // parts, timeit(lambda: ms[0](xs[0]))))
    def both():
        cur = torch.cuda.current_stream()
        for s in streams: s.wait_stream(cur)
        for mm, xx, s in zip(ms, xs, streams):
            with torch.cuda.stream(s):
                mm(xx)
        for s in streams: cur.wait_stream(s)
    for _ in range(3): both()
    print('%d streams, batch %d each: %.3f ms' % (parts, B // parts, timeit(both)))
