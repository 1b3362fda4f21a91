"""Generates tests/golden/*.npz by importing the reference (read-only, /root/reference) on the CPU.

Run once in the build container:  python tests/golden/make_golden.py
The reference imports cv2 and torchvision at module scope (utils/utils.py:13,19) although the path under
test never calls cv2 and calls torchvision only for ops.boxes.nms (utils/utils.py:627); neither package is
installed here, so empty stand-in modules satisfy the imports and `nms` is provided by the pure-torch
restatement below (torchvision's CPU algorithm).  Everything else in the fixtures - conv blocks, Focus,
BottleneckCSP, SPP, upsample+Concat, Detect decode in half precision, whole graphs, the stage list of
non_max_suppression, the filter-pruning channel plan - is computed by the reference's own code.
Only data (inputs, parameters, outputs) is written; no reference source text is stored.
"""
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    sys.dont_write_bytecode = True
    cv2 = types.ModuleType('cv2')
    cv2.setNumThreads = lambda n: None
    sys.modules['cv2'] = cv2
    tv, tvm, tvo, tvb = (types.ModuleType(n) for n in ('torchvision', 'torchvision.models', 'torchvision.ops', 'torchvision.ops.boxes'))
    tv.models, tv.ops, tvo.boxes = tvm, tvo, tvb
    tvb.nms = tvo.nms = nms_restated
    for n, m in (('torchvision', tv), ('torchvision.models', tvm), ('torchvision.ops', tvo), ('torchvision.ops.boxes', tvb)):
        sys.modules[n] = m
    import matplotlib
    matplotlib.use('Agg')
    sys.path.insert(0, REF)
    sys.path.insert(1, os.path.join(REF, 'filter_pruning'))
    os.chdir(REF)
    with contextlib.redirect_stdout(io.StringIO()):
        import models.common as common
        import models.yolo as yolo
        import utils.utils as uu
        import utils.torch_utils as tu
    return common, yolo, uu, tu


def nms_restated(boxes, scores, iou_threshold):
    """Greedy NMS with torchvision's CPU semantics (stable descending order, fp32 IoU compared in double)."""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.int64)
    order = torch.sort(scores, descending=True, stable=True).indices
    b = boxes[order].numpy().astype(np.float32)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    dead = np.zeros(n, bool)
    keep = []
    for i in range(n):
        if dead[i]:
            continue
        keep.append(i)
        w = np.maximum(np.float32(0), np.minimum(b[i, 2], b[i + 1:, 2]) - np.maximum(b[i, 0], b[i + 1:, 0]))
        h = np.maximum(np.float32(0), np.minimum(b[i, 3], b[i + 1:, 3]) - np.maximum(b[i, 1], b[i + 1:, 1]))
        inter = w * h
        with np.errstate(divide='ignore', invalid='ignore'):
            ovr = inter / (area[i] + area[i + 1:] - inter)
        dead[i + 1:] |= ovr.astype(np.float64) > iou_threshold
    return order[torch.as_tensor(keep, dtype=torch.int64)]


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def randomize_bn(module, seed):
    """Give every BatchNorm non-trivial statistics and affine parameters (deterministic, by state_dict key order)."""
    g = torch.Generator().manual_seed(seed)
    sd = module.state_dict()
    for k in sorted(sd):
        v = sd[k]
        if k.endswith('running_mean'):
            v.copy_(torch.randn(v.shape, generator=g) * 0.2)
        elif k.endswith('running_var'):
            v.copy_(torch.rand(v.shape, generator=g) * 1.5 + 0.5)
        elif k.endswith('bn.weight'):
            v.copy_(torch.rand(v.shape, generator=g) + 0.5)
        elif k.endswith('bn.bias'):
            v.copy_(torch.randn(v.shape, generator=g) * 0.2)


def save(name, **arrays):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **{k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()})
    print('%-28s %8.1f KB' % (name + '.npz', os.path.getsize(path) / 1024))


def sd_arrays(module, prefix='model.0.'):
    return {'sd.' + prefix + k: v for k, v in module.state_dict().items() if 'num_batches_tracked' not in k}


def main():
    common, yolo, uu, tu = import_reference()
    torch.set_grad_enabled(False)

    # ---------------------------------------------------------------- (i) blocks
    def block(name, ctor, cin, hw=(16, 16), seed=0, fuse=False, batch=2):
        torch.manual_seed(seed)
        m = ctor()
        tu.initialize_weights(m)
        randomize_bn(m, seed + 100)
        m.eval()
        if fuse:
            for c in m.modules():
                if type(c) is common.Conv:
                    c.conv = tu.fuse_conv_and_bn(c.conv, c.bn)
                    c.bn = None
                    c.forward = c.fuseforward
        x = torch.randn(batch, cin, *hw, generator=torch.Generator().manual_seed(seed + 1))
        y = m(x)
        save('block_' + name, x=x, y=y, **sd_arrays(m))

    block('conv_k1s1', lambda: common.Conv(16, 24, 1, 1), 16)
    block('conv_k3s1', lambda: common.Conv(16, 32, 3, 1), 16)
    block('conv_k3s2', lambda: common.Conv(24, 40, 3, 2), 24, hw=(16, 12))
    block('conv_k3s2_fused', lambda: common.Conv(24, 40, 3, 2), 24, hw=(16, 12), fuse=True)
    block('conv_k3s1_odd', lambda: common.Conv(13, 25, 3, 1), 13, hw=(9, 7))
    block('focus', lambda: common.Focus(3, 32, 3), 3, hw=(32, 24))
    block('bottleneck_add', lambda: common.Bottleneck(32, 32, True, e=1.0), 32)
    block('bottleneck_noadd', lambda: common.Bottleneck(32, 32, False, e=1.0), 32)
    block('csp', lambda: common.BottleneckCSP(32, 32, n=2, shortcut=True), 32)
    block('csp_noshortcut', lambda: common.BottleneckCSP(48, 32, n=1, shortcut=False), 48)
    block('csp_fused', lambda: common.BottleneckCSP(32, 32, n=1, shortcut=True), 32, fuse=True)
    block('spp', lambda: common.SPP(32, 32, (5, 9, 13)), 32, hw=(20, 20))

    for mode in ('nearest', 'bilinear'):
        g = torch.Generator().manual_seed(7)
        a, b = torch.randn(2, 16, 6, 5, generator=g), torch.randn(2, 24, 12, 10, generator=g)
        up = nn.Upsample(None, 2, mode)
        y = common.Concat(1)([up(a), b])
        save('block_upsample_%s_concat' % mode, a=a, b=b, y=y)

    # ---------------------------------------------------------------- (ii) Detect decode in half precision
    torch.manual_seed(3)
    anchors = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]
    for nc, sizes, tag in ((80, ((8, 12), (4, 6), (2, 3)), 'nc80_64x96'), (5, ((16, 16), (8, 8), (4, 4)), 'nc5_128x128'),
                           (1, ((5, 7), (3, 4), (2, 2)), 'nc1_odd')):
        det = yolo.Detect(nc, anchors, ch=(8, 8, 8))
        det.stride = torch.tensor([8., 16., 32.])
        det.m = nn.ModuleList([nn.Identity() for _ in range(3)])  # feed raw head values straight into the decode
        det.eval().half()
        g = torch.Generator().manual_seed(nc)
        raw = [(torch.randn(2, 3 * (nc + 5), ny, nx, generator=g) * 3.0).half() for ny, nx in sizes]
        z, xs = det([t.clone() for t in raw])
        save('decode_' + tag, z=z, anchor_grid=det.anchor_grid.float(), stride=det.stride,
             **{'raw%d' % i: t for i, t in enumerate(raw)}, **{'x%d' % i: t for i, t in enumerate(xs)})

    # ---------------------------------------------------------------- (iii) non_max_suppression
    def synth_pred(seed, B, N, nc, obj_hi=0.9, cls_hi=0.9, frac=0.3, cluster=40, span=320.0, quant=None):
        g = torch.Generator().manual_seed(seed)
        centers = torch.rand(B, cluster, 2, generator=g) * span
        pick = torch.randint(0, cluster, (B, N), generator=g)
        cxy = torch.gather(centers, 1, pick[..., None].expand(B, N, 2)) + torch.randn(B, N, 2, generator=g) * 6
        wh = torch.exp(torch.randn(B, N, 2, generator=g) * 0.4 + 3.2)
        obj = torch.rand(B, N, 1, generator=g) * obj_hi * (torch.rand(B, N, 1, generator=g) < frac)
        cls = torch.rand(B, N, nc, generator=g) ** 4 * cls_hi
        p = torch.cat((cxy, wh, obj, cls), 2)
        if quant:
            p[..., 4:] = (p[..., 4:] * quant).round() / quant  # few distinct scores -> many exact ties
        return p.float()

    cases = {
        'multilabel': dict(pred=synth_pred(1, 2, 1500, 8), conf=0.001, iou=0.65),
        'conf01': dict(pred=synth_pred(2, 2, 1500, 8), conf=0.1, iou=0.6),
        'ties': dict(pred=synth_pred(3, 2, 1200, 4, quant=8), conf=0.05, iou=0.5),
        'agnostic': dict(pred=synth_pred(4, 1, 1000, 6), conf=0.01, iou=0.45, agnostic=True),
        'classes': dict(pred=synth_pred(5, 2, 1000, 6), conf=0.01, iou=0.6, classes=[1, 4]),
        'over300': dict(pred=synth_pred(6, 1, 4000, 3, frac=0.9, cluster=2000, span=3000.0), conf=0.05, iou=0.3),
        'nc1': dict(pred=synth_pred(7, 2, 800, 1), conf=0.05, iou=0.6),
        'empty': dict(pred=synth_pred(8, 2, 300, 4, obj_hi=0.0005), conf=0.001, iou=0.6),
        'half_input': dict(pred=synth_pred(9, 2, 1500, 8).half(), conf=0.001, iou=0.65),
    }
    single = torch.zeros(1, 50, 7)
    single[0, 17] = torch.tensor([100., 120., 40., 60., 0.9, 0.2, 0.8])
    cases['single'] = dict(pred=single, conf=0.1, iou=0.6)
    for name, c in cases.items():
        out = uu.non_max_suppression(c['pred'].clone(), c['conf'], c['iou'], merge=False, classes=c.get('classes'),
                                     agnostic=c.get('agnostic', False))
        B = len(out)
        det = np.zeros((B, 300, 6), np.float32)
        cnt = np.zeros(B, np.int32)
        for b, o in enumerate(out):
            if o is not None:
                cnt[b] = o.shape[0]
                det[b, :o.shape[0]] = o.numpy()
        save('nms_' + name, pred=c['pred'], conf=c['conf'], iou=c['iou'], agnostic=int(c.get('agnostic', False)),
             classes=np.asarray(c.get('classes') or [], np.int64), det=det, count=cnt)
    c = cases['conf01']
    out = uu.non_max_suppression(c['pred'].clone(), c['conf'], c['iou'], merge=True)
    det = np.zeros((len(out), 300, 6), np.float32)
    cnt = np.zeros(len(out), np.int32)
    for b, o in enumerate(out):
        if o is not None:
            cnt[b] = o.shape[0]
            det[b, :o.shape[0]] = o.numpy()
    save('nms_merge', pred=c['pred'], conf=c['conf'], iou=c['iou'], det=det, count=cnt)

    # ---------------------------------------------------------------- (iv)+(v) whole graphs
    meta = {}
    for cfg, nc in (('yolov5s', None), ('yolov5m', None), ('yolov5l', None), ('yolov5x', None), ('yolov5s_ori', None)):
        torch.manual_seed(0)
        model = quiet(yolo.Model, 'models/%s.yaml' % cfg)
        model.eval()
        sd = model.state_dict()
        meta[cfg] = dict(
            keys=np.array([k for k in sd if 'num_batches_tracked' not in k]),
            shapes=np.array([';'.join(map(str, sd[k].shape)) for k in sd if 'num_batches_tracked' not in k]),
            save=np.asarray(model.save), stride=model.stride.numpy(),
            anchors=model.model[-1].anchors.numpy(), anchor_grid=model.model[-1].anchor_grid.numpy(),
            checksum=np.float64(sum(float(v.double().abs().sum()) for k, v in sd.items() if v.dtype.is_floating_point)))
        if cfg in ('yolov5s', 'yolov5s_ori'):
            randomize_bn(model, 11)
            extra = {}
            for tag, shape in (('a', (2, 3, 64, 64)), ('b', (1, 3, 128, 96))):
                x = torch.rand(*shape, generator=torch.Generator().manual_seed(5))
                z, xs = model(x)
                extra.update({'x_' + tag: x, 'z_' + tag: z, **{'x%d_%s' % (i, tag): t for i, t in enumerate(xs)}})
            fused = quiet(model.fuse)
            zf, _ = fused(extra['x_a'])
            extra['z_a_fused'] = zf
            save('e2e_' + cfg, **extra)
    for cfg, m in meta.items():
        save('graph_' + cfg, **m)

    # ---------------------------------------------------------------- (v) filter-pruned channel plan
    try:
        import importlib.util
        spec = importlib.util.spec_from_file_location('ref_testtrain_consts', os.path.join(REF, 'filter_pruning/unittest/TestTrain.py'))
        src = open(spec.origin).read()
        start = src.index('a0=[')
        a0 = eval(src[start + 3:src.index(']', start) + 1])  # the 68 per-layer rates of TestTrain.py:72-81
        from torch_pruning.pruning import filter_pruning
        torch.manual_seed(0)
        model = quiet(yolo.Model, 'models/yolov5s_ori.yaml').eval()
        torch.set_grad_enabled(True)
        torch.manual_seed(0)
        pruned = quiet(filter_pruning, model, 320, a0)
        torch.set_grad_enabled(False)
        sd = pruned.state_dict()
        keys = [k for k in sd if 'num_batches_tracked' not in k]
        y = pruned.eval()(torch.zeros(1, 3, 64, 64))[0]
        save('graph_yolov5s_ori_pruned_a0', keys=np.array(keys), shapes=np.array([';'.join(map(str, sd[k].shape)) for k in keys]),
             rates=np.asarray(a0), out_shape=np.asarray(y.shape))
    except Exception as e:  # recorded, not fatal: the plan fixture is optional input for the pruned-model tests
        print('pruned plan NOT generated:', repr(e))


if __name__ == '__main__':
    main()
