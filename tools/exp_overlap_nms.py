"""Experiment: forward(i+1) on the main stream beside NMS(i) on a second stream."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import util
from yolo_amd import nms as hip_nms

torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
m = util.build_model('yolov5s'); util.calibrate(m, size=640); m = util.quiet(m.fuse); m = m.eval().half().to(dev)
x = torch.rand(32, 3, 640, 640).half().to(dev)
for _ in range(3):
    z, _ = m(x); hip_nms.nms_padded(z, 0.001, 0.65, check_overflow=False)
def serial(n):
    for _ in range(n):
        z, _ = m(x)
        hip_nms.nms_padded(z, 0.001, 0.65, check_overflow=False)
side = torch.cuda.Stream()
def overlapped(n):
    main = torch.cuda.current_stream()
    for _ in range(n):
        z, _ = m(x)
        ev = torch.cuda.Event(); ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            z.record_stream(side)
            hip_nms.nms_padded(z, 0.001, 0.65, check_overflow=False)
    main.wait_stream(side)
def timeit(fn, n=30):
    fn(3); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(n); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print('forward only      %.3f ms' % timeit(lambda n: [m(x) for _ in range(n)]))
print('serial            %.3f ms/step' % timeit(serial))
print('NMS on 2nd stream %.3f ms/step' % timeit(overlapped))
