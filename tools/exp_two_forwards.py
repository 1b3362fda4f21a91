"""Experiment: do D independent batch-32 steps (forward + NMS) in flight on D streams (D replicas of the model, each with
its own launch plan, activation buffers and NMS workspace) finish sooner than the same steps back to back on one stream?
The small kernels of the 40x40 and 20x20 stages leave compute units idle; another step's kernels can use them.
usage: exp_two_forwards.py [model batch]"""
import os, sys, copy, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import bench
from yolo_amd import engine, nms as hip_nms

torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
name = sys.argv[1] if len(sys.argv) > 1 else 'yolov5s'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
m, _, _ = bench.build_model(name, dev)
x = torch.rand(B, 3, 640, 640).half().to(dev)
for _ in range(3):
    z, _ = m(x)
D = 4
reps = [m] + [copy.deepcopy(m) for _ in range(D - 1)]
xs = [x] + [x.clone() for _ in range(D - 1)]
posts = []
for mm, xx in zip(reps, xs):
    engine.set_output_ring(mm, 2)
    for _ in range(3):
        mm(xx)
    # max_candidates above the built-in array's capacity: every replica gets a workspace of its own
    posts.append(hip_nms.PaddedNms(*z.shape, z.dtype, dev, conf_thres=0.001, iou_thres=0.65, ring=2, max_candidates=65537))
assert not posts[0].builtin
streams = [torch.cuda.Stream() for _ in range(D)]


def run(n, depth, nms):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        k = i % depth
        with torch.cuda.stream(streams[k]):
            zz, _ = reps[k](xs[k])
            if nms:
                posts[k](zz)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for nms in (False, True):
    for _ in range(2):
        print('%s batch %d, %s: ms per step with 1 / 2 / 3 / 4 steps in flight: %s' % (
            name, B, 'forward + NMS' if nms else 'forward only', ' / '.join('%.4f' % run(120, d, nms) for d in (1, 2, 3, 4))), flush=True)
