"""Diagnostic: how far the HIP fp16 path and torch-ROCm's own fp16 kernels are from the half-precision oracle on the
calibrated synthetic yolov5s at 640x640 (max / mean / percentiles of |d| on the raw head maps and on z)."""
import os, sys
import torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import util
from oracle import y5ref
from yolo_amd import engine
torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
gain = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
m = util.build_model('yolov5s'); util.calibrate(m, seed=1, size=640, head_gain=gain)
m = util.quiet(m.fuse).eval().half().to(dev)
x = torch.rand(2, 3, 640, 640, generator=torch.Generator().manual_seed(1)).half()
z, xs = m(x.to(dev))
with engine.disabled():
    zt, xt = m(x.to(dev))
cfg = yaml.safe_load(open(os.path.join(util.CFG_DIR, 'yolov5s.yaml')))
sd = {k: v.float().cpu() for k, v in m.state_dict().items()}
(zr, xr), feats = y5ref.forward(cfg, sd, x.float(), precision='half', strides=m.stride, return_features=True)
def stats(tag, a, b):
    d = (a.float().cpu() - b.float().cpu()).abs().flatten()
    q = torch.quantile(d[:: max(1, d.numel() // 2000000)], torch.tensor([0.5, 0.99, 0.999, 0.9999]))
    print('%-22s max %.4g mean %.3g p50 %.3g p99 %.3g p99.9 %.3g p99.99 %.3g exact %.4f  max|ref| %.3g' % (
        tag, d.max(), d.mean(), q[0], q[1], q[2], q[3], float((d == 0).float().mean()), float(b.float().abs().max())))
for i in range(3):
    stats('hip   vs oracle x%d' % i, xs[i], xr[i])
    stats('torch vs oracle x%d' % i, xt[i], xr[i])
    stats('hip   vs torch  x%d' % i, xs[i], xt[i])
stats('hip   vs oracle z', z, zr)
stats('torch vs oracle z', zt, zr)
