"""Experiment: one plan of batch 32 on one stream versus sub-batches on concurrent streams (forward only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import util
from yolo_amd import engine

dev = torch.device('cuda:0')
model = util.build_model('yolov5s')
util.calibrate(model)
model = model.fuse().half().eval().to(dev)
B = 32
x = torch.rand(B, 3, 640, 640, device=dev).half()

def bench(parts, iters=20):
    xs = list(x.chunk(parts))
    plans = [engine.Plan(model, xs[0]) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    def once():
        for p, s, xi in zip(plans, streams, xs):
            with torch.cuda.stream(s):
                p.run(xi)
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        once()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print('parts %d: %.3f ms per batch of %d  (%.0f img/s forward only)' % (parts, dt * 1e3, B, B / dt), flush=True)
    for p in plans:
        p.close()

with torch.no_grad():
    for parts in (1, 2, 4):
        bench(parts)
