"""Engine forward against torch's own kernels on the same modules over a sweep of batch sizes and image shapes
(ragged tiles, Detect levels whose runs are not 16-byte aligned, pruned channel counts).  Prints one line per case."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import util
from yolo_amd import engine

torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
bad = 0
for cfg, pruned in (('yolov5s', False), ('yolov5s_ori', True), ('yolov5m', False)):
    model = util.build_model(cfg)
    if pruned:
        model = util.prune_like(model)
    util.randomize_bn(model, 7)
    model = util.quiet(model.fuse).eval().half().to(dev)
    for B, H, W in ((1, 320, 256), (3, 416, 416), (2, 608, 352), (5, 224, 640), (1, 64, 64), (7, 160, 96)):
        x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(B * H + W)).half().to(dev)
        z, xs = model(x)
        eng = engine.engine_of(model)
        with engine.disabled():
            zt, xt = model(x)
        d = (z.float() - zt.float()).abs()
        tol = 4e-3 * zt.float().abs() + 4e-3 * zt.float().abs().max()
        ok = bool((d <= tol).all()) and all(a.shape == b.shape for a, b in zip(xs, xt))
        dx = max(float((a.float() - b.float()).abs().max()) for a, b in zip(xs, xt))
        bad += not ok
        print('%-12s B%d %dx%d: z max|d| %.4g (max|z| %.4g) raw max|d| %.4g exact %.3f %s' % (
            cfg + ('/pruned' if pruned else ''), B, H, W, float(d.max()), float(zt.float().abs().max()), dx,
            float((d == 0).float().mean()), 'ok' if ok else 'MISMATCH'))
print('%d mismatching cases' % bad)
sys.exit(1 if bad else 0)
