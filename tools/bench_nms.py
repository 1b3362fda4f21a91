"""Time the NMS stages on the bench workload's predictions. usage: bench_nms.py [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import util
from yolo_amd import nms as hip_nms

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
model = util.build_model('yolov5s')
util.calibrate(model)
model = model.fuse().half().eval().to(dev)
x = torch.rand(B, 3, 640, 640, device=dev).half()
with torch.no_grad():
    z = model(x)[0]
for _ in range(3):
    out = hip_nms.nms_padded(z, 0.001, 0.65, check_overflow=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    out = hip_nms.nms_padded(z, 0.001, 0.65, check_overflow=False)
e1.record()
torch.cuda.synchronize()
print('nms_padded: %.1f us per call, kept %.1f per image' % (e0.elapsed_time(e1) / 20 * 1e3, out[1].float().mean().item()))
