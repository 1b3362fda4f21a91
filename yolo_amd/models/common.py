"""Building blocks of the YOLOv5-v2.0 graph (mirror of the reference's models/common.py).

The classes keep the reference's constructor arguments and sub-module attribute names
(conv, bn, act, cv1..cv4, m) because state_dict keys, pickled checkpoints and the
filter-pruning tools address them by name, and `Conv.conv` / `Conv.bn` stay real
nn.Conv2d / nn.BatchNorm2d modules that those tools mutate in place.  Their forward()
methods are the plain torch formulation; on a GPU in inference mode Model.forward
bypasses them and runs the whole graph through the HIP engine (yolo_amd/engine.py), which
reads shapes and parameters from these live modules.
"""
import math

import torch
import torch.nn as nn


def autopad(k, p=None):
    """'same' padding for an odd kernel: k//2, per dimension for tuples (reference: models/common.py:6-10)."""
    if p is not None:
        return p
    return k // 2 if isinstance(k, int) else [v // 2 for v in k]


class Conv(nn.Module):
    """conv2d(bias=False) -> BatchNorm2d -> LeakyReLU(0.1) (reference: models/common.py:18-30)."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p), groups=g, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.act = nn.LeakyReLU(0.1, inplace=True) if act else nn.Identity()

    def forward(self, x):
        return self.act(self.bn(self.conv(x)))

    def fuseforward(self, x):
        """Forward after Model.fuse() folded the BatchNorm into self.conv (reference: models/common.py:29-30)."""
        return self.act(self.conv(x))


def DWConv(c1, c2, k=1, s=1, act=True):
    """Depthwise-style Conv with groups = gcd(c1, c2) (reference: models/common.py:13-15)."""
    return Conv(c1, c2, k, s, g=math.gcd(c1, c2), act=act)


class Bottleneck(nn.Module):
    """1x1 Conv -> 3x3 Conv with an optional identity shortcut (reference: models/common.py:33-43)."""

    def __init__(self, c1, c2, shortcut=True, g=1, e=0.5):
        super().__init__()
        hidden = int(c2 * e)
        self.cv1 = Conv(c1, hidden, 1, 1)
        self.cv2 = Conv(hidden, c2, 3, 1, g=g)
        self.add = shortcut and c1 == c2

    def forward(self, x):
        y = self.cv2(self.cv1(x))
        return x + y if self.add else y


class BottleneckCSP(nn.Module):
    """Cross-stage-partial block (reference: models/common.py:46-62).

    y1 = cv3(m(cv1(x))) and y2 = cv2(x) are plain 1x1 convs; BatchNorm + LeakyReLU act on their
    channel concat before cv4.  Sub-modules are created in the reference's order so that a seeded
    construction draws the same random initial weights.
    """

    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5):
        super().__init__()
        hidden = int(c2 * e)
        self.cv1 = Conv(c1, hidden, 1, 1)
        self.cv2 = nn.Conv2d(c1, hidden, 1, 1, bias=False)
        self.cv3 = nn.Conv2d(hidden, hidden, 1, 1, bias=False)
        self.cv4 = Conv(2 * hidden, c2, 1, 1)
        self.bn = nn.BatchNorm2d(2 * hidden)
        self.act = nn.LeakyReLU(0.1, inplace=True)
        self.m = nn.Sequential(*[Bottleneck(hidden, hidden, shortcut, g, e=1.0) for _ in range(n)])

    def forward(self, x):
        y1 = self.cv3(self.m(self.cv1(x)))
        y2 = self.cv2(x)
        return self.cv4(self.act(self.bn(torch.cat((y1, y2), dim=1))))


class SPP(nn.Module):
    """Spatial pyramid pooling: cv2(cat[x, maxpool_k(x) for k in (5, 9, 13)]) with x = cv1(x)
    (reference: models/common.py:65-76)."""

    def __init__(self, c1, c2, k=(5, 9, 13)):
        super().__init__()
        hidden = c1 // 2
        self.cv1 = Conv(c1, hidden, 1, 1)
        self.cv2 = Conv(hidden * (len(k) + 1), c2, 1, 1)
        self.m = nn.ModuleList([nn.MaxPool2d(kernel_size=v, stride=1, padding=v // 2) for v in k])

    def forward(self, x):
        x = self.cv1(x)
        return self.cv2(torch.cat([x] + [pool(x) for pool in self.m], 1))


class Focus(nn.Module):
    """Space-to-depth then Conv (reference: models/common.py:79-86).  Output channel of the gather is
    q*C + c with q enumerating (row parity, col parity) as (0,0), (1,0), (0,1), (1,1)."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True):
        super().__init__()
        self.conv = Conv(c1 * 4, c2, k, s, p, g, act)

    @staticmethod
    def space_to_depth(x):
        return torch.cat([x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]], 1)

    def forward(self, x):
        return self.conv(self.space_to_depth(x))


class Concat(nn.Module):
    """torch.cat of a list of tensors along `dimension` (reference: models/common.py:89-96)."""

    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension

    def forward(self, x):
        return torch.cat(x, self.d)


class Flatten(nn.Module):
    """(B,C,1,1) -> (B,C) (reference: models/common.py:99-103)."""

    @staticmethod
    def forward(x):
        return x.view(x.size(0), -1)


class Classify(nn.Module):
    """Global-average-pool classification head (reference: models/common.py:106-116); torch ops only."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1):
        super().__init__()
        self.aap = nn.AdaptiveAvgPool2d(1)
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p), groups=g, bias=False)
        self.flat = Flatten()

    def forward(self, x):
        pooled = torch.cat([self.aap(y) for y in (x if isinstance(x, list) else [x])], 1)
        return self.flat(self.conv(pooled))
