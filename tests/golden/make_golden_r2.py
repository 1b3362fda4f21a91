"""Second batch of fixtures from the imported reference (same import recipe as make_golden.py, which this reuses):

  tta_yolov5s.npz       Model.forward(x, augment=True) (models/yolo.py:89-103), fp32, seeded yolov5s
  ensemble_yolov5s.npz  Ensemble.forward of two seeded, fused models (models/experimental.py:113-125), fp32
  halfblock_<name>.npz  the block fixtures' modules under .half() on the CPU: the reference's half-precision
                        evaluation tensor by tensor, which the oracle's precision='half' mode restates
  bf16block_<name>.npz  the same under .bfloat16() (stored as fp32 values that are exact bfloat16 numbers)
  decode_bf16_64x96.npz Detect.forward under .bfloat16() on the CPU (rounding order of the bf16 decode)
Run once in the build container:  python tests/golden/make_golden_r2.py
Only data is written.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402


def main():
    common, yolo, uu, tu = mg.import_reference()
    import models.experimental as experimental
    torch.set_grad_enabled(False)

    # ---- test-time augmentation
    torch.manual_seed(0)
    model = mg.quiet(yolo.Model, 'models/yolov5s.yaml').eval()
    mg.randomize_bn(model, 11)
    x = torch.rand(2, 3, 96, 128, generator=torch.Generator().manual_seed(7))
    y, none = model(x, augment=True)
    assert none is None
    mg.save('tta_yolov5s', x=x, y=y)

    # ---- ensemble of two models (what attempt_load builds from two checkpoints): fused, eval, mean of the outputs
    ens = experimental.Ensemble()
    for seed in (0, 1):
        torch.manual_seed(seed)
        m = mg.quiet(yolo.Model, 'models/yolov5s.yaml')
        mg.randomize_bn(m, 11 + seed)
        ens.append(mg.quiet(m.float().fuse).eval())
    x = torch.rand(1, 3, 64, 96, generator=torch.Generator().manual_seed(8))
    y, none = ens(x)
    assert none is None
    mg.save('ensemble_yolov5s', x=x, y=y, y0=ens[0](x)[0], y1=ens[1](x)[0])

    # ---- blocks in half precision on the CPU (same constructors, seeds and inputs as make_golden.block)
    def halfblock(name, ctor, cin, hw=(16, 16), seed=0, fuse=False, batch=2):
        torch.manual_seed(seed)
        m = ctor()
        tu.initialize_weights(m)
        mg.randomize_bn(m, seed + 1)
        m.eval()
        if fuse:
            for c in m.modules():
                if type(c) is common.Conv:
                    c.conv = tu.fuse_conv_and_bn(c.conv, c.bn)
                    c.bn = None
                    c.forward = c.fuseforward
        x = torch.randn(batch, cin, *hw, generator=torch.Generator().manual_seed(seed + 100))
        mh = m.half()
        yh = mh(x.half())
        mg.save('halfblock_' + name, x=x.half(), y=yh, **mg.sd_arrays(mh))

    def bf16block(name, ctor, cin, hw=(16, 16), seed=0, batch=2):
        torch.manual_seed(seed)
        m = ctor()
        tu.initialize_weights(m)
        mg.randomize_bn(m, seed + 1)
        m.eval()
        x = torch.randn(batch, cin, *hw, generator=torch.Generator().manual_seed(seed + 100))
        mb = m.bfloat16()
        yb = mb(x.bfloat16())
        mg.save('bf16block_' + name, x=x.bfloat16().float(), y=yb.float(), **{k: v.float() for k, v in mg.sd_arrays(mb).items()})

    bf16block('conv_k3s1', lambda: common.Conv(16, 32, 3, 1), 16)
    bf16block('csp', lambda: common.BottleneckCSP(32, 32, n=2, shortcut=True), 32)
    bf16block('spp', lambda: common.SPP(32, 32, (5, 9, 13)), 32, hw=(20, 20))
    bf16block('focus', lambda: common.Focus(3, 32, 3), 3, hw=(32, 32))

    halfblock('conv_k3s1', lambda: common.Conv(16, 32, 3, 1), 16)
    halfblock('conv_k3s2_fused', lambda: common.Conv(24, 40, 3, 2), 24, fuse=True)
    halfblock('bottleneck_add', lambda: common.Bottleneck(32, 32, True, e=1.0), 32)
    halfblock('csp', lambda: common.BottleneckCSP(32, 32, n=2, shortcut=True), 32)
    halfblock('spp', lambda: common.SPP(32, 32, (5, 9, 13)), 32, hw=(20, 20))
    halfblock('focus', lambda: common.Focus(3, 32, 3), 3, hw=(32, 32))

    # ---- Detect decode under bfloat16
    torch.manual_seed(0)
    det = yolo.Detect(80, [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]], [8, 8, 8])
    det.stride = torch.tensor([8., 16., 32.])
    det.eval()
    det = det.bfloat16()
    for c in det.m:  # identity-like 1x1 convs are not needed: feed the decode through hooks on the conv outputs instead
        pass
    g = torch.Generator().manual_seed(3)
    raws = [(torch.randn(2, 255, 64 // s, 96 // s, generator=g) * 3).bfloat16() for s in (8, 16, 32)]

    class Pass(torch.nn.Module):
        def forward(self, t):
            return t
    det.m = torch.nn.ModuleList([Pass() for _ in range(3)])
    z, xs = det([r.clone() for r in raws])
    arrays = {'z': z.float(), 'stride': det.stride, 'anchor_grid': det.anchor_grid.float()}
    for i in range(3):
        arrays['raw%d' % i] = raws[i].float()
        arrays['x%d' % i] = xs[i].float()
    mg.save('decode_bf16_64x96', **arrays)


if __name__ == '__main__':
    main()
