"""Per-op device time table of one forward plan (hipEvent pairs around every launch) + NMS stage time."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import util
from yolo_amd import nms as hip_nms

ap = argparse.ArgumentParser()
ap.add_argument('--model', default='yolov5s'); ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--size', type=int, default=640); ap.add_argument('--passes', type=int, default=5)
ap.add_argument('--nofuse', action='store_true'); ap.add_argument('--pruned', action='store_true'); ap.add_argument('--summary', action='store_true')
a = ap.parse_args()
torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
m = util.build_model(a.model)
if a.pruned:
    m = util.prune_like(m)
util.calibrate(m, size=a.size)
if not a.nofuse:
    m = util.quiet(m.fuse)
m = m.eval().half().to(dev)
x = torch.rand(a.batch, 3, a.size, a.size, generator=torch.Generator().manual_seed(1)).half().to(dev)
for _ in range(3):
    z, _ = m(x)
from yolo_amd import engine as _eng_mod
eng = _eng_mod.engine_of(m)
acc = None
for _ in range(a.passes):
    rows = eng.profile(x)
    if acc is None: acc = rows
    else:
        for p, r in zip(acc, rows): p['ms'] += r['ms']
tot = 0
for r in acc:
    r['ms'] /= a.passes; tot += r['ms']
    if not a.summary: print('%3d %-8s %8.1f us %8.1f TF/s %8.0f GB/s  %s' % (r['op'], r['kind'], r['ms'] * 1e3, r['flops'] / r['ms'] / 1e9 if r['ms'] else 0, r['bytes'] / r['ms'] / 1e6 if r['ms'] else 0, r['label']))
conv=[r for r in acc if r['kind'] in ('conv','focus_conv','detect')]
cf=sum(r['flops'] for r in conv); cm=sum(r['ms'] for r in conv)
print('%s batch %d: plan total %.3f ms, conv %.3f ms, %.1f GFLOP/img, conv aggregate %.1f TFLOP/s' % (a.model, a.batch, tot, cm, cf/a.batch/1e9, cf/cm/1e9))
for r in sorted(conv, key=lambda r:-r['ms'])[:8]: print('   top: %7.1f us %7.1f TF/s %s' % (r['ms']*1e3, r['flops']/r['ms']/1e9, r['label']))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): z, _ = m(x)
e1.record(); torch.cuda.synchronize()
print('forward wall (10 runs) %.3f ms each' % (e0.elapsed_time(e1) / 10))
e0.record()
for _ in range(10): hip_nms.nms_padded(z, 0.001, 0.65, check_overflow=False)
e1.record(); torch.cuda.synchronize()
print('nms %.3f ms each' % (e0.elapsed_time(e1) / 10))
