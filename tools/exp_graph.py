"""Does replaying the forward as a captured graph beat issuing its launches one by one?  Device time of 50 forwards either
way (events), and the host's wall time per call.  usage: exp_graph.py [--model yolov5s --batch 32]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch

ap = argparse.ArgumentParser()
ap.add_argument('--model', default='yolov5s'); ap.add_argument('--batch', type=int, default=32)
args = ap.parse_args()
import bench
from yolo_amd import engine
torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
model, _, _ = bench.build_model(args.model, dev)
x = torch.rand(args.batch, 3, 640, 640).half().to(dev)
engine.set_output_ring(model, 1)
for _ in range(5):
    z, _ = model(x)
torch.cuda.synchronize()


def timed(fn, n=50):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    host = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, host


print('eager   : %.4f ms device per forward, %.4f ms host per call' % timed(lambda: model(x)), flush=True)
ref = z.clone()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    model(x)
    with torch.cuda.graph(g, stream=s):
        zg, _ = model(x)
torch.cuda.current_stream().wait_stream(s)
zg.zero_()
g.replay()
torch.cuda.synchronize()
print('graph output equals eager:', torch.equal(zg, ref), flush=True)
print('graph   : %.4f ms device per forward, %.4f ms host per call' % timed(g.replay), flush=True)
print('eager   : %.4f ms device per forward, %.4f ms host per call' % timed(lambda: model(x)), flush=True)
