"""Device time of BottleneckCSP(64, 64, n=1) at 160x160: the one-kernel form against the layer-by-layer plan."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import util
from yolo_amd import engine
from yolo_amd.models import common
from yolo_amd.utils import torch_utils

torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(0)
m = common.BottleneckCSP(64, 64, n=1, shortcut=True)
util.randomize_bn(m, 1)
m.eval()
for c in m.modules():
    if type(c) is common.Conv:
        c.conv = torch_utils.fuse_conv_and_bn(c.conv, c.bn); c.bn = None; c.forward = c.fuseforward
m = m.half().to(dev)
x = torch.randn(B, 64, 160, 160).half().to(dev)
for mode in ('force', '0'):
    os.environ['Y5_CSP_FUSE'] = mode
    plan = engine.Plan(m, x, block=True)
    for _ in range(3):
        plan.run_block(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import ctypes as C
    from yolo_amd import _lib
    arr = (C.c_void_p * 1)(x.data_ptr())
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        plan.lib.y5_plan_run(plan.handle, arr, 1, _lib.current_stream())
    e1.record(); torch.cuda.synchronize()
    print(mode, '%.1f us per block (incl. NCHW->NHWC pack)' % (e0.elapsed_time(e1) / 20 * 1e3), plan.labels)
