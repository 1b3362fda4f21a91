"""Experiment: the filter-pruned yolov5s_ori (widths such as 13, 25, 55, 119, 249) with its private buffers padded to
multiples of 8 channels (what the engine does) against padded up to the next regular width (32, 64, 128, ... when that is
within a third, else the next multiple of 32), so that the kernels written for regular widths apply."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import util
from yolo_amd import engine

torch.set_grad_enabled(False)
dev = torch.device('cuda:0')


def regular(c):
    if c <= 16:
        return (c + 7) // 8 * 8
    for w in (32, 64, 128, 256, 512, 1024):
        if c <= w <= c * 1.34:
            return w
    return (c + 31) // 32 * 32


def build():
    m = util.build_model('yolov5s_ori')
    m = util.prune_like(m)
    util.randomize_bn(m, 11)
    m.eval()
    util.quiet(m.fuse)
    return m.half().to(dev)


x = torch.rand(32, 3, 640, 640, generator=torch.Generator().manual_seed(1)).half().to(dev)
outs = {}
for name, fn in (('multiples of 8', engine.pad8), ('regular widths', regular)):
    engine.pad8 = fn
    m = build()
    for _ in range(3):
        z, _ = m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        m(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 30 * 1e3
    plan = next(iter(engine.engine_of(m).plans.values()))
    print('%-16s forward %.3f ms, %d launches' % (name, ms, plan.lib.y5_plan_num_ops(plan.handle)), flush=True)
    outs[name] = z.float().cpu()
a, b = outs['multiples of 8'], outs['regular widths']
d = (a - b).abs()
print('max |dz| %.4g of max |z| %.4g; identical %.4f' % (d.max(), a.abs().max(), (d == 0).float().mean()))
