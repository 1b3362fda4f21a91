"""Detect head, graph executor and yaml graph builder (mirror of the reference's models/yolo.py).

Model.forward keeps the reference contract: eval mode returns (cat(z, 1), [x_i]) with
z rows ordered level-major, then anchor, then y, then x; train mode returns [x_i].
When the call is eligible (see yolo_amd.engine.eligible) the whole graph - packed NHWC fp16
convolutions with folded BatchNorm, SPP, upsample/concat, the head convs and the box decode -
runs as HIP kernels; otherwise the modules are walked with torch ops exactly like the reference.
"""
import math
from copy import deepcopy
from pathlib import Path

import torch
import torch.nn as nn

from yolo_amd.models.common import (Bottleneck, BottleneckCSP, Concat, Conv, DWConv, Focus, SPP, Classify, Flatten,
                                    autopad)  # noqa: F401  (re-exported like the reference's star import)
from yolo_amd.utils import torch_utils
from yolo_amd.utils.utils import check_anchor_order, make_divisible


class Detect(nn.Module):
    """Detection head: per level a biased 1x1 conv to na*(nc+5) channels, then box decode
    (reference: models/yolo.py:8-51)."""

    def __init__(self, nc=80, anchors=(), ch=()):
        super().__init__()
        self.stride = None                      # filled by Model.__init__ from a probe forward
        self.nc = nc
        self.no = nc + 5                        # outputs per anchor: cx, cy, w, h, obj, classes
        self.nl = len(anchors)
        self.na = len(anchors[0]) // 2
        self.grid = [torch.zeros(1)] * self.nl
        a = torch.tensor(anchors).float().view(self.nl, -1, 2)
        self.register_buffer('anchors', a)                                        # (nl, na, 2), in stride units after build
        self.register_buffer('anchor_grid', a.clone().view(self.nl, 1, -1, 1, 1, 2))  # (nl,1,na,1,1,2), pixels
        self.m = nn.ModuleList(nn.Conv2d(c, self.no * self.na, 1) for c in ch)
        self.export = False

    def forward(self, x):
        """x: list of nl feature maps (B, C_i, ny, nx).  The dtype order of the decode in half precision
        (sigmoid, *2, -0.5 in fp16; +grid, *stride in fp32; one rounding on store; wh entirely in fp16)
        is what the HIP decode kernel reproduces (reference: models/yolo.py:37-46)."""
        self.training |= self.export
        z = []
        for i in range(self.nl):
            x[i] = self.m[i](x[i])
            bs, _, ny, nx = x[i].shape
            x[i] = x[i].view(bs, self.na, self.no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
            if self.training:
                continue
            if self.grid[i].shape[2:4] != x[i].shape[2:4]:
                self.grid[i] = self._make_grid(nx, ny).to(x[i].device)
            y = x[i].sigmoid()
            y[..., 0:2] = (y[..., 0:2] * 2. - 0.5 + self.grid[i].to(x[i].device)) * self.stride[i]
            y[..., 2:4] = (y[..., 2:4] * 2) ** 2 * self.anchor_grid[i]
            z.append(y.view(bs, -1, self.no))
        return x if self.training else (torch.cat(z, 1), x)

    @staticmethod
    def _make_grid(nx=20, ny=20):
        yv, xv = torch.meshgrid([torch.arange(ny), torch.arange(nx)], indexing='ij')
        return torch.stack((xv, yv), 2).view((1, 1, ny, nx, 2)).float()


class Model(nn.Module):
    """YOLOv5 graph built from a yaml description (reference: models/yolo.py:54-166)."""

    def __init__(self, cfg='yolov5s.yaml', ch=3, nc=None):
        super().__init__()
        if isinstance(cfg, dict):
            self.yaml = cfg
        else:
            import yaml
            self.yaml_file = Path(cfg).name
            if not Path(cfg).is_file() and (Path(__file__).parent / self.yaml_file).is_file():
                cfg = str(Path(__file__).parent / self.yaml_file)  # graphs shipped with the package
            with open(cfg) as f:
                self.yaml = yaml.load(f, Loader=yaml.SafeLoader)
        if nc and nc != self.yaml['nc']:
            print('Overriding %s nc=%g with nc=%g' % (cfg, self.yaml['nc'], nc))
            self.yaml['nc'] = nc
        self.model, self.save = parse_model(deepcopy(self.yaml), ch=[ch])

        head = self.model[-1]
        if isinstance(head, Detect):
            probe = 128  # twice the smallest supported stride-32 map
            feats = self.forward(torch.zeros(1, ch, probe, probe))
            head.stride = torch.tensor([probe / f.shape[-2] for f in feats])
            head.anchors /= head.stride.view(-1, 1, 1)
            check_anchor_order(head)
            self.stride = head.stride
            self._initialize_biases()
        torch_utils.initialize_weights(self)
        self.info()
        print('')

    # ------------------------------------------------------------------ forward
    def forward(self, x, augment=False, profile=False):
        if not augment:
            return self.forward_once(x, profile)
        # test-time augmentation: original, flipped+0.83x, 0.67x (reference: models/yolo.py:89-103)
        if isinstance(x, torch.Tensor) and x.dtype is torch.uint8:
            # the flipped / rescaled views below are new tensors: a from_hwc batch would lose its channel-order marking
            raise TypeError('augment=True needs a floating-point batch (img.half() / 255.0); uint8 input is for plain forwards')
        height_width = x.shape[-2:]
        scales = (0.83, 0.67)
        views = (x, torch_utils.scale_img(x.flip(3), scales[0]), torch_utils.scale_img(x, scales[1]))
        y = [self.forward_once(v)[0] for v in views]
        y[1][..., :4] /= scales[0]
        y[1][..., 0] = height_width[1] - y[1][..., 0]
        y[2][..., :4] /= scales[1]
        return torch.cat(y, 1), None

    def forward_once(self, x, profile=False):
        from yolo_amd import engine
        if not profile:
            out = engine.try_run(self, x)
            if out is not None:
                return out
        if isinstance(x, torch.Tensor) and x.dtype is torch.uint8:  # same convenience off the engine: test.py:152-154
            if engine.input_layout(x) == engine._lib.IN_U8_HWC_BGR:  # yolo_amd.preprocess.from_hwc(..., bgr=True): utils/datasets.py:140
                x = x.flip(1)
            x = x.to(next(self.parameters()).dtype) / 255.0
            if not profile:
                out = engine.try_run(self, x)
                if out is not None:
                    return out
        return self._forward_modules(x, profile)

    def _forward_modules(self, x, profile=False):
        """Walk the module list with torch ops (reference: models/yolo.py:107-130)."""
        kept, times = [], []
        for m in self.model:
            if m.f != -1:
                x = kept[m.f] if isinstance(m.f, int) else [x if j == -1 else kept[j] for j in m.f]
            if profile:
                t = torch_utils.time_synchronized()
                for _ in range(10):
                    _ = m(x)
                times.append((torch_utils.time_synchronized() - t) * 100)
                print('%10.0f%10.1fms %-40s' % (m.np, times[-1], m.type))
            x = m(x)
            kept.append(x if m.i in self.save else None)
        if profile:
            print('%.1fms total' % sum(times))
        return x

    # ------------------------------------------------------------------ setup helpers
    def _initialize_biases(self, cf=None):
        """Detect bias prior: obj ~ 8 objects per 640 image, cls ~ 0.6/(nc-0.99) (reference: models/yolo.py:132-142)."""
        head = self.model[-1]
        for conv, s in zip(head.m, head.stride):
            b = conv.bias.view(head.na, -1)
            with torch.no_grad():
                b[:, 4] += math.log(8 / (640 / s) ** 2)
                b[:, 5:] += math.log(0.6 / (head.nc - 0.99)) if cf is None else torch.log(cf / cf.sum())
            conv.bias = torch.nn.Parameter(b.view(-1), requires_grad=True)

    def fuse(self):
        """Fold BatchNorm into every `Conv` (exact type match, so the BatchNorm that BottleneckCSP applies
        to its concat stays live) and switch those modules to fuseforward (reference: models/yolo.py:155-163)."""
        print('Fusing layers... ', end='')
        for m in self.model.modules():
            if type(m) is Conv:
                m.conv = torch_utils.fuse_conv_and_bn(m.conv, m.bn)
                m.bn = None
                m.forward = m.fuseforward
        self.info()
        return self

    def info(self):
        torch_utils.model_info(self)


_BLOCKS = {'Conv': Conv, 'Bottleneck': Bottleneck, 'BottleneckCSP': BottleneckCSP, 'SPP': SPP, 'Focus': Focus,
           'Concat': Concat, 'DWConv': DWConv, 'Detect': Detect, 'nn.Upsample': nn.Upsample,
           'nn.Conv2d': nn.Conv2d, 'nn.BatchNorm2d': nn.BatchNorm2d}
_WIDTH_SCALED = (nn.Conv2d, Conv, Bottleneck, SPP, DWConv, Focus, BottleneckCSP)


def _resolve(token, names):
    """yaml argument strings name a few graph-level values; everything else stays a literal."""
    if not isinstance(token, str):
        return token
    if token in names:
        return names[token]
    if token in ('None', 'True', 'False'):
        return {'None': None, 'True': True, 'False': False}[token]
    return token


def parse_model(d, ch):
    """yaml dict -> (nn.Sequential, sorted list of layer indices whose outputs later layers read)
    (reference: models/yolo.py:169-229).  Rules: repeats n>1 are scaled by depth_multiple (min 1),
    output widths by width_multiple rounded up to a multiple of 8, except the Detect width na*(nc+5)."""
    anchors, nc, gd, gw = d['anchors'], d['nc'], d['depth_multiple'], d['width_multiple']
    na = (len(anchors[0]) // 2) if isinstance(anchors, list) else anchors
    no = na * (nc + 5)
    names = {'nc': nc, 'anchors': anchors}

    layers, save, c2 = [], [], ch[-1]
    for i, (f, n, m, args) in enumerate(d['backbone'] + d['head']):
        block = _BLOCKS[m] if isinstance(m, str) else m
        args = [_resolve(a, names) for a in args]
        n = max(round(n * gd), 1) if n > 1 else n
        if block in _WIDTH_SCALED:
            c1, c2 = ch[f], args[0]
            c2 = make_divisible(c2 * gw, 8) if c2 != no else c2
            args = [c1, c2, *args[1:]]
            if block is BottleneckCSP:
                args.insert(2, n)
                n = 1
        elif block is nn.BatchNorm2d:
            args = [ch[f]]
        elif block is Concat:
            c2 = sum(ch[-1 if j == -1 else j + 1] for j in f)
        elif block is Detect:
            args.append([ch[j + 1] for j in f])
            if isinstance(args[1], int):
                args[1] = [list(range(args[1] * 2))] * len(f)
        else:
            c2 = ch[f]

        module = nn.Sequential(*[block(*args) for _ in range(n)]) if n > 1 else block(*args)
        module.i, module.f = i, f
        module.type = 'nn.' + block.__name__ if block.__module__.startswith('torch.nn') else block.__module__.split('yolo_amd.')[-1] + '.' + block.__name__
        module.np = sum(p.numel() for p in module.parameters())
        save.extend(j % i for j in ([f] if isinstance(f, int) else f) if j != -1)
        layers.append(module)
        ch.append(c2)
    return nn.Sequential(*layers), sorted(save)
