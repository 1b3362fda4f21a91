"""Time the steps after NMS on a batch of 32 padded detection blocks: scale_coords + clip, and label matching."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yolo_amd import metrics, nms as hip_nms

dev = torch.device('cuda:0')
B = 32
g = torch.Generator().manual_seed(0)
det = torch.zeros(B, 300, 6)
det[..., :2] = torch.rand(B, 300, 2, generator=g) * 500
det[..., 2:4] = det[..., :2] + torch.rand(B, 300, 2, generator=g) * 140
det[..., 4] = torch.sort(torch.rand(B, 300, generator=g), descending=True).values
det[..., 5] = torch.randint(0, 80, (B, 300), generator=g).float()
count = torch.full((B,), 300, dtype=torch.int32)
nl = 40
tg = torch.cat([torch.cat((torch.full((nl, 1), float(b)), torch.randint(0, 80, (nl, 1), generator=g).float(),
                           torch.rand(nl, 2, generator=g) * 0.8 + 0.1, torch.rand(nl, 2, generator=g) * 0.2 + 0.02), 1) for b in range(B)], 0)
det, count, tgd = det.to(dev), count.to(dev), tg.to(dev)
iouv = torch.linspace(0.5, 0.95, 10).to(dev)
shapes = [(480, 640, 3)] * B


def timeit(fn, n=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print('match_predictions (32 images x 300 detections x 40 labels, 10 IoU thresholds): %.1f us per batch'
      % timeit(lambda: metrics.match_predictions(det, count, tgd, 640, 640, iouv)))
print('scale_detections (32 x 300): %.1f us per batch (includes the host-side parameter upload)'
      % timeit(lambda: hip_nms.scale_detections(det, count, (640, 640), shapes)))
