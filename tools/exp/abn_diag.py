"""Diagnostic: adaptive-BN statistics of the HIP path against the fp32 oracle and against the oracle restated with fp16
activations, per model and input size (worst error over all BatchNorm channels)."""
import os, sys, torch, yaml
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import util
from oracle import y5ref
from yolo_amd.adaptive_bn import Recalibrator
def cmp(a, b, tag):
    wm = wv = 0
    for k in a:
        if k.endswith('running_mean'):
            kv = k.replace('running_mean', 'running_var')
            std = b[kv].double().sqrt()
            wm = max(wm, ((a[k].cpu().double() - b[k].double()).abs() / std).max().item())
            wv = max(wv, ((a[kv].cpu().double() - b[kv].double()).abs() / b[kv].double()).max().item())
    print('  %-22s worst mean err/std %.4f  worst var rel err %.4f' % (tag, wm, wv))
for cfg, shape in (('yolov5s', (8, 3, 128, 96)), ('yolov5m', (8, 3, 128, 128)), ('yolov5s', (16, 3, 320, 320))):
    model = util.build_model(cfg); util.randomize_bn(model, 5)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d): m.momentum = 1.0
    g = torch.Generator().manual_seed(1)
    batches = [torch.randint(0, 256, shape, generator=g, dtype=torch.uint8) for _ in range(2)]
    cfgd = yaml.safe_load(open(os.path.join(util.CFG_DIR, cfg + '.yaml')))
    ref = y5ref.adaptive_bn(cfgd, model.state_dict(), batches, momentum=1.0)
    refh = y5ref.adaptive_bn(cfgd, model.state_dict(), batches, momentum=1.0, precision='half')
    model = model.cuda().train(); rec = Recalibrator(model)
    for b in batches: rec.step(b.cuda())
    rec.finish(); sd = model.state_dict()
    print(cfg, shape)
    cmp(sd, ref, 'hip vs fp32 oracle'); cmp(sd, refh, 'hip vs half oracle'); cmp(refh, ref, 'half vs fp32 oracle')
