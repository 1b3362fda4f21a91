"""Host-side helpers of the inference path (mirror of the reference's utils/torch_utils.py).

Only the functions the conv -> Detect -> NMS path and its callers touch are provided:
device selection, synchronized timing, BatchNorm defaults, Conv+BN folding, the
test-time-augmentation image scaler and the parameter summary.  Training-only helpers
of the reference (ModelEMA, unstructured prune, classifier loading) are not part of
this path and are absent on purpose.
"""
import math
import os
import time
from copy import deepcopy

import torch
import torch.nn as nn
import torch.nn.functional as F


def init_seeds(seed=0):
    """Seed torch (reference: utils/torch_utils.py:13-22; the cudnn flags have no ROCm meaning)."""
    torch.manual_seed(seed)


def select_device(device='', apex=False, batch_size=None):
    """'cpu', '' (auto) or a comma list of GPU ordinals -> torch.device (reference: utils/torch_utils.py:25-49).

    Must run before anything initialises the GPU runtime because it narrows the visible devices
    through the environment, exactly like the reference does.
    """
    want_cpu = device.lower() == 'cpu'
    if device and not want_cpu:
        os.environ['CUDA_VISIBLE_DEVICES'] = device
        os.environ.setdefault('HIP_VISIBLE_DEVICES', device)
        assert torch.cuda.is_available(), 'GPU unavailable, invalid device %s requested' % device
    use_gpu = (not want_cpu) and torch.cuda.is_available()
    if not use_gpu:
        print('Using CPU\n')
        return torch.device('cpu')
    n = torch.cuda.device_count()
    if n > 1 and batch_size:
        assert batch_size % n == 0, 'batch-size %g not multiple of GPU count %g' % (batch_size, n)
    for i in range(n):
        p = torch.cuda.get_device_properties(i)
        print('%sdevice%g %s (%dMB)' % ('Using GPU ' if i == 0 else '          ', i, p.name, p.total_memory // 2 ** 20))
    print('')
    return torch.device('cuda:0')


def time_synchronized():
    """Wall clock after draining the device queue (reference: utils/torch_utils.py:52-54)."""
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return time.time()


def is_parallel(model):
    return type(model) in (nn.parallel.DataParallel, nn.parallel.DistributedDataParallel)


def initialize_weights(model):
    """BatchNorm eps 1e-3 / momentum 0.03, in-place activations (reference: utils/torch_utils.py:62-71)."""
    for m in model.modules():
        kind = type(m)
        if kind is nn.BatchNorm2d:
            m.eps, m.momentum = 1e-3, 0.03
        elif kind in (nn.LeakyReLU, nn.ReLU, nn.ReLU6):
            m.inplace = True


def bn_affine(bn, dtype=torch.float32):
    """Per-channel (scale, shift) of an eval-mode BatchNorm2d: y = x*scale + shift, computed in `dtype`."""
    var = bn.running_var.detach().to(dtype)
    mean = bn.running_mean.detach().to(dtype)
    gamma = torch.ones_like(var) if bn.weight is None else bn.weight.detach().to(dtype)
    beta = torch.zeros_like(var) if bn.bias is None else bn.bias.detach().to(dtype)
    scale = gamma / torch.sqrt(var + bn.eps)
    return scale, beta - mean * scale


def fuse_conv_and_bn(conv, bn):
    """Fold an eval-mode BatchNorm2d into the preceding bias-free conv (reference: utils/torch_utils.py:99-120).

    W' = diag(gamma / sqrt(var + eps)) W,  b' = beta - gamma * mean / sqrt(var + eps) (+ folded conv bias),
    evaluated in the dtype the module currently holds (fp32 at load time in the reference,
    models/experimental.py:133).  Like the reference, `groups` is not propagated to the new conv.
    """
    with torch.no_grad():
        fused = nn.Conv2d(conv.in_channels, conv.out_channels, kernel_size=conv.kernel_size, stride=conv.stride,
                          padding=conv.padding, bias=True).to(conv.weight.device)
        g = bn.weight.div(torch.sqrt(bn.eps + bn.running_var))
        w2d = conv.weight.clone().view(conv.out_channels, -1)
        fused.weight.copy_(torch.mm(torch.diag(g), w2d).view(fused.weight.size()))
        b0 = torch.zeros(conv.weight.size(0), device=conv.weight.device) if conv.bias is None else conv.bias
        shift = bn.bias - bn.weight.mul(bn.running_mean).div(torch.sqrt(bn.running_var + bn.eps))
        fused.bias.copy_(torch.mm(torch.diag(g), b0.reshape(-1, 1)).reshape(-1) + shift)
    return fused


def model_info(model, verbose=False):
    """One-line parameter summary (reference: utils/torch_utils.py:123-141; the optional thop FLOP line is dropped)."""
    params = list(model.parameters())
    n_p = sum(p.numel() for p in params)
    n_g = sum(p.numel() for p in params if p.requires_grad)
    if verbose:
        for i, (name, p) in enumerate(model.named_parameters()):
            print('%5g %40s %9s %12g %20s' % (i, name, p.requires_grad, p.numel(), list(p.shape)))
    print('Model Summary: %g layers, %g parameters, %g gradients' % (len(params), n_p, n_g))


def scale_img(img, ratio=1.0, same_shape=False):
    """Bilinear rescale of a (B,3,H,W) batch, padded up to a multiple of 32 with 0.447
    (reference: utils/torch_utils.py:165-173; used by test-time augmentation, models/yolo.py:93-96)."""
    h, w = img.shape[2:]
    new = (int(h * ratio), int(w * ratio))
    img = F.interpolate(img, size=new, mode='bilinear', align_corners=False)
    if not same_shape:
        h, w = [math.ceil(v * ratio / 32) * 32 for v in (h, w)]
    return F.pad(img, [0, w - new[1], 0, h - new[0]], value=0.447)


def copy_attr(a, b, include=(), exclude=()):
    for k, v in b.__dict__.items():
        if (len(include) and k not in include) or k.startswith('_') or k in exclude:
            continue
        setattr(a, k, v)
