"""Forward time of yolov5s at batch 32 with Detect's first levels on the plan's side stream (Y5_SIDE_STREAM=2) against the
one-stream plan, alternated in one process.  Measured: 1.326 against 1.313-1.323 ms (no gain worth a second stream; the
same with the persistent Detect kernel limited to 128, 64 or 32 workgroups).  usage: exp_side_detect.py [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import bench
from yolo_amd import engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda:0')
model = bench.build_model('yolov5s', dev)[0]
x = torch.rand(B, 3, 640, 640, device=dev).half()
outs = {}
for rnd in range(2):
    for mode in ('0', '2'):
        os.environ['Y5_SIDE_STREAM'] = mode
        engine._ENGINES.pop(model, None)  # a fresh engine records its plan under the current setting
        with torch.no_grad():
            for _ in range(3):
                y = model(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                y = model(x)
            e1.record(); torch.cuda.synchronize()
        outs[mode] = y[0].clone()
        print('Y5_SIDE_STREAM=%s  forward %.4f ms' % (mode, e0.elapsed_time(e1) / 30))
print('outputs identical:', bool(torch.equal(outs['0'], outs['2'])))
