"""Checkpoint loader and model ensemble (mirror of the reference's models/experimental.py:113-141).

The experimental blocks of that file (CrossConv, C3, Sum, GhostConv, GhostBottleneck, MixConv2d)
appear in none of the shipped yolov5{s,m,l,x}/yolov5s_ori graphs and are not part of this path.
"""
import torch
import torch.nn as nn


class Ensemble(nn.ModuleList):
    """Mean of the inference outputs of several models (reference: models/experimental.py:113-125)."""

    def forward(self, x, augment=False):
        y = torch.stack([module(x, augment)[0] for module in self]).mean(0)
        return y, None


def attempt_load(weights, map_location=None):
    """Load one checkpoint or a list of them as fp32, BatchNorm-fused, eval-mode models
    (reference: models/experimental.py:128-141).  Checkpoints pickle whole modules under the names
    models.yolo.* / models.common.*, so yolo_amd.dropin.install() must have aliased those module
    paths first; files are local (the reference's download step needs a network)."""
    from yolo_amd import dropin
    dropin.install()
    models = Ensemble()
    for w in weights if isinstance(weights, list) else [weights]:
        ckpt = torch.load(w, map_location=map_location, weights_only=False)
        models.append(ckpt['model'].float().fuse().eval())
    if len(models) == 1:
        return models[-1]
    print('Ensemble created with %s\n' % weights)
    for k in ('names', 'stride'):
        setattr(models, k, getattr(models[-1], k))
    return models
