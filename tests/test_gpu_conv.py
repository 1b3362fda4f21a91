"""HIP convolution path against the CPU oracle (half-precision restatement, itself pinned to the reference's modules
under .half() by tests/test_oracle_golden.py), against the reference's own fp32 block outputs, and, for shapes that
reach every tile variant of the kernel, against a torch fp32 convolution of the same fp16 operands.

Tolerance for a single convolution (close_ulp): the kernel accumulates in fp32 in a different order than the checker
and rounds to fp16 at the same points, so an intermediate that falls next to a rounding boundary may come out one fp16
step apart: |y - ref| <= 2 ulp16(max(|ref|, magnitudes of the values added on the way: bias / shift / residual
operands)) + the fp32 summation-order bound 2^-18 * sum|x||w|.  Multi-layer blocks (close): a first-layer step is
carried through the later layers with their gain, so the step is taken at the tensor's scale:
|y - ref| <= 2e-3*|ref| + 2e-3*max|ref|; in practice > 99 % of the elements are bit-identical."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

import util
from oracle import y5ref
from yolo_amd import _lib, engine

pytestmark = pytest.mark.gpu
OPS16 = y5ref._Ops('half')


def close(got, ref, rel=2e-3):
    got, ref = got.float().cpu(), ref.float().cpu()
    tol = rel * ref.abs() + rel * ref.abs().max()
    bad = (got - ref).abs() > tol
    assert not bool(bad.any()), 'max |diff| %g (max |ref| %g), %d elements out of tolerance' % (
        float((got - ref).abs().max()), float(ref.abs().max()), int(bad.sum()))
    return float(((got - ref) == 0).float().mean())


def ulp16(t):
    """Spacing of fp16 numbers at |t| (fp32 in / out; the subnormal spacing below 2^-14)."""
    a = t.abs().clamp_min(2.0 ** -14)
    return torch.pow(2.0, torch.floor(torch.log2(a)) - 10)


def close_ulp(got, ref, mags=(), slack=None, ulps=2.0):
    """|got - ref| <= ulps * ulp16(max(|ref|, *mags)) + slack, elementwise."""
    got, ref = got.float().cpu(), ref.float().cpu()
    m = ref.abs()
    for t in mags:
        m = torch.maximum(m, t.float().cpu().abs())
    tol = ulps * ulp16(m)
    if slack is not None:
        tol = tol + slack.float().cpu()
    d = (got - ref).abs()
    bad = d > tol
    assert not bool(bad.any()), 'worst %.2f of the tolerance, %d elements out' % (float((d / tol).max()), int(bad.sum()))
    return float((d == 0).float().mean())


@pytest.mark.parametrize('name', util.BLOCKS)
def test_block_against_reference_fp32_output(dev, name):
    """The HIP block (fp16 operands, fp32 accumulation) against the fixture's own `y`: what the reference's modules
    computed in fp32.  The difference is the precision of the path, not of the kernels: inputs and weights rounded to
    fp16 (2^-11 relative each) and one fp16 rounding per stored tensor.  Bound: 3e-3*|y| + 3e-3*max|y| (measured below
    half of it on every block)."""
    m, x, y = util.block_from_fixture(name)
    got = engine.run_block(m.half().to(dev), x.half().to(dev)).float().cpu()
    d = (got - y).abs()
    tol = 3e-3 * y.abs() + 3e-3 * y.abs().max()
    assert got.shape == y.shape and bool((d <= tol).all()), 'worst %.2f of the tolerance' % float((d / tol).max())


@pytest.mark.parametrize('name', util.BLOCKS)
def test_block_against_oracle(dev, name):
    m, x, _ = util.block_from_fixture(name)
    sd = {'model.0.' + k: v for k, v in m.state_dict().items()}
    xh = x.half()
    if name.startswith('conv'):
        ref = y5ref.conv_block(OPS16, sd, 'model.0', xh.float(), 2 if 's2' in name else 1)
    elif name == 'focus':
        ref = y5ref.conv_block(OPS16, sd, 'model.0.conv', y5ref.focus_gather(xh.float()))
    elif name.startswith('bottleneck'):
        ref = y5ref.bottleneck(OPS16, sd, 'model.0', xh.float(), 'noadd' not in name)
    elif name.startswith('csp'):
        ref = y5ref.bottleneck_csp(OPS16, sd, 'model.0', xh.float(), len(m.m), 'noshortcut' not in name)
    else:
        ref = y5ref.spp(OPS16, sd, 'model.0', xh.float())
    got = engine.run_block(m.half().to(dev), xh.to(dev))
    assert got.shape == ref.shape
    exact = close(got, ref)
    assert exact > 0.98


def run_conv(dev, x, w, bias=None, scale=None, shift=None, slope=None, res=None, stride=1, variant=0):
    """Call y5_conv_f16 directly. x (B,C,H,W) fp16, w (Co,C,k,k) fp16 -> (B,Co,Ho,Wo) fp16.
    With an explicit kernel variant, returns None when that variant does not take the shape."""
    lib = _lib.lib()
    B, Ci, H, W = x.shape
    Co, _, k, _ = w.shape
    cip, cop = (Ci + 7) // 8 * 8, (Co + 7) // 8 * 8
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    xn = torch.zeros(B, H, W, cip, dtype=torch.float16, device=dev)
    xn[..., :Ci] = x.permute(0, 2, 3, 1)
    kk = k * k * cip
    kpad, rows = (kk + 63) // 64 * 64, (Co + 255) // 256 * 256
    wp = torch.zeros(rows, kpad, dtype=torch.float16, device=dev)
    wv = torch.zeros(Co, k, k, cip, dtype=torch.float16, device=dev)
    wv[..., :Ci] = w.permute(0, 2, 3, 1)
    wp[:Co, :kk] = wv.reshape(Co, kk)
    y = torch.full((B, Ho, Wo, cop), float('nan'), dtype=torch.float16, device=dev)
    keep = []

    def vec(t):
        if t is None:
            return None
        v = torch.zeros(cop, dtype=torch.float32, device=dev)
        v[:Co] = t.float()
        keep.append(v)
        return v.data_ptr()
    d = _lib.ConvDesc()
    d.x, d.w, d.y = xn.data_ptr(), wp.data_ptr(), y.data_ptr()
    d.bias, d.scale, d.shift = vec(bias), vec(scale), vec(shift)
    if res is not None:
        rn = torch.zeros(B, Ho, Wo, cop, dtype=torch.float16, device=dev)
        rn[..., :Co] = res.permute(0, 2, 3, 1)
        d.res, d.res_ld = rn.data_ptr(), cop
    d.batch, d.h, d.w_in, d.cin, d.x_ld = B, H, W, cip, cip
    d.ho, d.wo, d.cout, d.y_ld = Ho, Wo, cop, cop
    d.ksize, d.stride, d.pad, d.kpad, d.w_rows = k, stride, k // 2, kpad, rows
    d.act, d.slope = (_lib.ACT_LEAKY, slope) if slope is not None else (_lib.ACT_NONE, 0.0)
    d.variant = variant
    rc = lib.y5_conv_f16(C.byref(d), _lib.current_stream())
    if variant and rc == _lib.EINVAL:
        return None
    _lib.check(rc)
    torch.cuda.synchronize()
    assert not bool(torch.isnan(y).any()), 'conv left output elements unwritten'
    assert bool((y[..., Co:] == 0).all()), 'padding channels must be zero'
    return y[..., :Co].permute(0, 3, 1, 2)


# (B, Cin, Cout, H, W, k, stride): chosen to reach every tile / K-step / addressing variant of the kernel
SHAPES = [
    (2, 64, 128, 40, 40, 1, 1),    # 64x128 tile, BK 64, direct
    (8, 128, 128, 80, 80, 1, 1),   # 128x128 tile
    (2, 32, 32, 48, 40, 1, 1),     # 128x32 tile, BK 32 (K = 32)
    (4, 16, 32, 64, 64, 3, 1),     # K = 144 -> BK 32
    (2, 32, 64, 50, 38, 3, 2),     # 64x64 tile, stride 2, odd sizes
    (16, 32, 64, 80, 80, 3, 2),    # 128x64 tile
    (1, 256, 256, 20, 20, 3, 1),   # long K
    (2, 128, 255, 20, 12, 1, 1),   # Detect-like width, non-square
    (3, 13, 25, 9, 7, 3, 1),       # pruned-style odd channels (padded to 16 / 32)
    (2, 96, 192, 16, 16, 1, 2),    # 1x1 stride 2 (not direct)
    (16, 128, 128, 40, 40, 3, 1),  # 128x128 tile, 3x3
]


@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_conv_kernel_against_torch_fp32(dev, shape):
    B, Ci, Co, H, W, k, s = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Ci, H, W, generator=g).half().to(dev)
    w = (torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5).half().to(dev)
    bias = torch.randn(Co, generator=g).to(dev)
    scale, shift = (torch.rand(Co, generator=g) + 0.5).to(dev), torch.randn(Co, generator=g).to(dev)
    acc = F.conv2d(x.float(), w.float(), None, s, k // 2)
    order = F.conv2d(x.float().abs(), w.float().abs(), None, s, k // 2) * 2.0 ** -18  # fp32 summation-order bound
    bv, sv, hv = bias.view(1, -1, 1, 1), scale.view(1, -1, 1, 1), shift.view(1, -1, 1, 1)
    # bias + leaky (the fused-BatchNorm form)
    ref = F.leaky_relu((acc + bv).half().float(), 0.1).half()
    close_ulp(run_conv(dev, x, w, bias=bias, slope=0.1, stride=s), ref, mags=(acc, bv.expand_as(acc)), slack=order)
    # folded BatchNorm + leaky + residual (the unfused form inside a Bottleneck)
    res = torch.randn(ref.shape, generator=g).half().to(dev)
    r2 = (acc.half().float() * sv + hv).half().float()
    pre = F.leaky_relu(r2, 0.1).half().float()
    r2 = (pre + res.float()).half()
    close_ulp(run_conv(dev, x, w, scale=scale, shift=shift, slope=0.1, res=res, stride=s), r2,
              mags=(acc * sv, hv.expand_as(acc), pre, res), slack=order * sv.abs())
    # plain (no post-op)
    close_ulp(run_conv(dev, x, w, stride=s), acc.half(), slack=order)


# Shapes for the explicit-variant sweep: ragged pixel counts (a partial last tile), K that is not a power of two,
# channel counts below and above one tile.
VARIANT_SHAPES = [
    (3, 32, 32, 48, 41, 1, 1),
    (2, 64, 128, 40, 40, 1, 1),
    (1, 200, 72, 33, 17, 1, 1),
    (5, 128, 255, 20, 12, 1, 1),
    (9, 256, 256, 20, 20, 1, 1),
    (2, 32, 64, 24, 41, 3, 1),
    (1, 64, 136, 19, 23, 3, 1),
    (2, 24, 48, 30, 30, 3, 2),
    (2, 32, 64, 32, 64, 3, 2),     # the persistent stride-2 kernel's shape class (output 16 x 32)
    (3, 32, 64, 16, 96, 3, 2),
    (3, 32, 64, 50, 38, 3, 2),
    (2, 64, 64, 33, 47, 3, 1),
    (2, 32, 24, 64, 64, 3, 1),
    (2, 160, 320, 12, 20, 3, 1),
    (2, 80, 96, 20, 24, 3, 1),
    (1, 40, 72, 17, 33, 3, 2),
    # more tiles than resident workgroups: the persistent variants walk several tiles each
    (32, 64, 64, 80, 80, 3, 1),
    (16, 128, 200, 56, 72, 1, 1),
    (24, 32, 48, 96, 64, 3, 2),
    # the persistent 64 -> 128 stride-2 kernel's shape class (output height % 4 == 0, width % 16 == 0): two tiles; 432 tiles
    (2, 64, 128, 8, 32, 3, 2),
    (9, 64, 128, 48, 64, 3, 2),
    # the persistent 1x1 kernel's channel classes; pixel counts that are not whole tiles, fewer and more tiles than workgroups
    (3, 256, 256, 21, 23, 1, 1),
    (32, 256, 256, 40, 40, 1, 1),
    (2, 128, 128, 37, 5, 1, 1),
    (32, 128, 128, 40, 40, 1, 1),
    (5, 256, 128, 33, 31, 1, 1),
    (5, 128, 256, 33, 31, 1, 1),
    # the same kernel on widths inside a power-of-two layout (yolov5m: 96, 192; yolov5x: 160): pieces and channel blocks past
    # the real width are skipped
    (3, 160, 160, 21, 23, 1, 1),
    (16, 160, 160, 40, 40, 1, 1),
    (2, 192, 192, 37, 5, 1, 1),
    (5, 96, 96, 33, 31, 1, 1),
    (3, 192, 96, 17, 19, 1, 1),
    (2, 320, 160, 33, 31, 1, 1),
    (9, 384, 192, 20, 20, 1, 1),
]
INNER_1X1 = {(160, 160): True, (192, 192): True, (96, 96): True, (192, 96): False, (320, 160): False, (384, 192): False}  # -> both tile sizes?


@pytest.mark.parametrize('shape', VARIANT_SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_every_kernel_variant_that_takes_the_shape(dev, shape):
    """y5_conv_desc.variant selects a tile configuration / kernel family explicitly (the engine's autotuner does
    so per layer): each one that accepts the shape must give the same result as the rule-chosen kernel's contract."""
    B, Ci, Co, H, W, k, s = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Ci, H, W, generator=g).half().to(dev)
    w = (torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5).half().to(dev)
    bias = torch.randn(Co, generator=g).to(dev)
    scale, shift = (torch.rand(Co, generator=g) + 0.5).to(dev), torch.randn(Co, generator=g).to(dev)
    acc = F.conv2d(x.float(), w.float(), None, s, k // 2)
    order = F.conv2d(x.float().abs(), w.float().abs(), None, s, k // 2) * 2.0 ** -18  # fp32 summation-order bound
    bv, sv, hv = bias.view(1, -1, 1, 1), scale.view(1, -1, 1, 1), shift.view(1, -1, 1, 1)
    ref_bias = F.leaky_relu((acc + bv).half().float(), 0.1).half()
    ref_bn = (acc.half().float() * sv + hv).half()
    res = torch.randn(ref_bn.shape, generator=g).half().to(dev)
    pre = F.leaky_relu(ref_bn.float(), 0.1).half().float()
    ref_res = (pre + res.float()).half()
    bn_mags = (acc * sv, hv.expand_as(acc))
    taken = 0
    for v in range(1, _lib.lib().y5_conv_num_variants() + 1):
        got = run_conv(dev, x, w, bias=bias, slope=0.1, stride=s, variant=v)
        if got is None:
            continue
        taken += 1
        close_ulp(got, ref_bias, mags=(acc, bv.expand_as(acc)), slack=order)
        close_ulp(run_conv(dev, x, w, scale=scale, shift=shift, stride=s, variant=v), ref_bn, mags=bn_mags, slack=order * sv.abs())
        close_ulp(run_conv(dev, x, w, stride=s, variant=v), acc.half(), slack=order)
        got = run_conv(dev, x, w, scale=scale, shift=shift, slope=0.1, res=res, stride=s, variant=v)
        if got is not None:  # some kernel families take no residual input
            close_ulp(got, ref_res, mags=bn_mags + (pre, res), slack=order * sv.abs())
    assert taken >= 10
    nph = _lib.lib().y5_conv_num_variants_phased()   # the phased kernels are the last ids and take every shape
    nv = _lib.lib().y5_conv_num_variants() - nph
    for v in range(nv + 1, nv + nph + 1):
        assert run_conv(dev, x, w, bias=bias, slope=0.1, stride=s, variant=v) is not None, 'the phased kernels take every 1x1 / 3x3 shape'
    if (Ci, Co, k, s) == (32, 64, 3, 2) and (H // 2) % 8 == 0 and (W // 2) % 16 == 0:
        assert run_conv(dev, x, w, bias=bias, slope=0.1, stride=s, variant=nv) is not None, 'the persistent stride-2 kernel takes this shape'
    if k == 1 and (Ci, Co) in INNER_1X1:
        assert run_conv(dev, x, w, bias=bias, slope=0.1, stride=s, variant=nv - 3) is not None, 'the persistent 1x1 kernel takes this shape'
        assert (run_conv(dev, x, w, bias=bias, slope=0.1, stride=s, variant=nv - 4) is not None) == INNER_1X1[(Ci, Co)]
    if k == 1 and Ci in (128, 256) and Co in (128, 256):
        assert run_conv(dev, x, w, bias=bias, slope=0.1, stride=s, variant=nv - 3) is not None, 'the persistent 1x1 kernel takes this shape'
        assert (run_conv(dev, x, w, bias=bias, slope=0.1, stride=s, variant=nv - 4) is not None) == ((Ci, Co) != (256, 128))
    if (Ci, Co, k, s) == (64, 128, 3, 2) and (H // 2) % 4 == 0 and (W // 2) % 16 == 0:
        assert run_conv(dev, x, w, bias=bias, slope=0.1, stride=s, variant=nv - 2) is not None, 'the persistent 64-channel stride-2 kernel takes this shape'


@pytest.mark.parametrize('shape', [(2, 64, 96, 32), (3, 42, 74, 32), (1, 128, 66, 48), (2, 36, 130, 80), (1, 640, 640, 32)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_focus_conv_from_the_nchw_image(dev, shape):
    """y5_focus_conv_f16 (Focus gather + 3x3 convolution in one kernel) against torch fp32 on the same fp16 operands:
    ragged tiles, 1..3 channel groups, both post-op forms, output into a wider buffer."""
    lib = _lib.lib()
    B, H, W, Co = shape
    g = torch.Generator().manual_seed(sum(shape))
    img = torch.randn(B, 3, H, W, generator=g).half().to(dev)
    w = (torch.randn(Co, 12, 3, 3, generator=g) / 108 ** 0.5).half().to(dev)
    bias = torch.randn(Co, generator=g).to(dev)
    scale, shift = (torch.rand(Co, generator=g) + 0.5).to(dev), torch.randn(Co, generator=g).to(dev)
    s2d = torch.cat([img[..., ::2, ::2], img[..., 1::2, ::2], img[..., ::2, 1::2], img[..., 1::2, 1::2]], 1)
    acc = F.conv2d(s2d.float(), w.float(), None, 1, 1)
    wp = torch.zeros(128, 192, dtype=torch.float16, device=dev)
    wv = torch.zeros(Co, 3, 3, 16, dtype=torch.float16, device=dev)
    wv[..., :12] = w.permute(0, 2, 3, 1)
    wp[:Co, :144] = wv.reshape(Co, 144)
    ld = Co + 8
    for mode in ('bias', 'bn', 'plain'):
        y = torch.full((B, H // 2, W // 2, ld), 7.0, dtype=torch.float16, device=dev)
        d = _lib.ConvDesc()
        d.x, d.w, d.y = img.data_ptr(), wp.data_ptr(), y.data_ptr()
        d.batch, d.h, d.w_in, d.cin = B, H, W, 3
        d.ho, d.wo, d.cout, d.y_ld = H // 2, W // 2, Co, ld
        d.ksize, d.stride, d.pad, d.kpad, d.w_rows = 3, 1, 1, 192, 128
        if mode == 'bias':
            d.bias, d.act, d.slope = bias.data_ptr(), _lib.ACT_LEAKY, 0.1
            ref = F.leaky_relu((acc + bias.view(1, -1, 1, 1)).half().float(), 0.1).half()
        elif mode == 'bn':
            d.scale, d.shift, d.act, d.slope = scale.data_ptr(), shift.data_ptr(), _lib.ACT_LEAKY, 0.1
            ref = (acc.half().float() * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).half()
            ref = F.leaky_relu(ref.float(), 0.1).half()
        else:
            ref = acc.half()
        _lib.check(lib.y5_focus_conv_f16(C.byref(d), _lib.current_stream()))
        torch.cuda.synchronize()
        assert bool((y[..., Co:] == 7).all()), 'wrote outside its channel slice'
        close(y[..., :Co].permute(0, 3, 1, 2), ref)


@pytest.mark.parametrize('shape', [(2, 64, 96, 32), (3, 42, 74, 48), (1, 160, 128, 80)], ids=lambda s: 'x'.join(map(str, s)))
def test_focus_conv_u8_equals_normalise_then_focus_conv(dev, shape):
    """y5_focus_conv_u8 folds img.half() / 255.0 into the stem kernel: bit-identical to normalising first."""
    from yolo_amd import preprocess
    lib = _lib.lib()
    B, H, W, Co = shape
    g = torch.Generator().manual_seed(sum(shape))
    img8 = torch.randint(0, 256, (B, 3, H, W), generator=g, dtype=torch.int32).to(torch.uint8).to(dev)
    img16 = preprocess.normalize(img8)
    wp = torch.zeros(128, 192, dtype=torch.float16, device=dev)
    wp[:Co, :144] = (torch.randn(Co, 144, generator=g) / 10).half().to(dev)
    wp.view(128, 12, 16)[:, :, 12:] = 0
    bias = torch.randn(Co, generator=g).to(dev)
    outs = []
    for img, fn in ((img16, lib.y5_focus_conv_f16), (img8, lib.y5_focus_conv_u8)):
        y = torch.zeros(B, H // 2, W // 2, Co, dtype=torch.float16, device=dev)
        d = _lib.ConvDesc()
        d.x, d.w, d.y, d.bias = img.data_ptr(), wp.data_ptr(), y.data_ptr(), bias.data_ptr()
        d.batch, d.h, d.w_in, d.cin = B, H, W, 3
        d.ho, d.wo, d.cout, d.y_ld = H // 2, W // 2, Co, Co
        d.ksize, d.stride, d.pad, d.kpad, d.w_rows = 3, 1, 1, 192, 128
        d.act, d.slope = _lib.ACT_LEAKY, 0.1
        _lib.check(fn(C.byref(d), _lib.current_stream()))
        outs.append(y)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and float(outs[0].abs().max()) > 0


@pytest.mark.parametrize('shape', [(2, 64, 96, 32), (3, 42, 74, 48), (1, 160, 128, 80), (2, 640, 640, 32)], ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('bgr', [True, False], ids=['bgr', 'rgb'])
def test_focus_conv_from_interleaved_uint8_equals_the_loaders_statements_first(dev, shape, bgr):
    """y5_focus_conv_u8_hwc reads (batch, h, w, 3) uint8 frames: bit-identical to `img[:, :, ::-1].transpose(2, 0, 1)` +
    np.ascontiguousarray (utils/datasets.py:140-141, done here with numpy on the host exactly as the loader does) followed
    by y5_focus_conv_u8."""
    import numpy as np
    lib = _lib.lib()
    B, H, W, Co = shape
    g = torch.Generator().manual_seed(sum(shape))
    frames = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.int32).to(torch.uint8).numpy()
    loader = np.stack([np.ascontiguousarray((f[:, :, ::-1] if bgr else f).transpose(2, 0, 1)) for f in frames])
    img_chw, img_hwc = torch.from_numpy(loader).to(dev), torch.from_numpy(frames).to(dev)
    wp = torch.zeros(128, 192, dtype=torch.float16, device=dev)
    wp[:Co, :144] = (torch.randn(Co, 144, generator=g) / 10).half().to(dev)
    wp.view(128, 12, 16)[:, :, 12:] = 0
    bias = torch.randn(Co, generator=g).to(dev)
    outs = []
    for img, hwc in ((img_chw, False), (img_hwc, True)):
        y = torch.zeros(B, H // 2, W // 2, Co, dtype=torch.float16, device=dev)
        d = _lib.ConvDesc()
        d.x, d.w, d.y, d.bias = img.data_ptr(), wp.data_ptr(), y.data_ptr(), bias.data_ptr()
        d.batch, d.h, d.w_in, d.cin = B, H, W, 3
        d.ho, d.wo, d.cout, d.y_ld = H // 2, W // 2, Co, Co
        d.ksize, d.stride, d.pad, d.kpad, d.w_rows = 3, 1, 1, 192, 128
        d.act, d.slope = _lib.ACT_LEAKY, 0.1
        if hwc:
            _lib.check(lib.y5_focus_conv_u8_hwc(C.byref(d), _lib.IN_U8_HWC_BGR if bgr else _lib.IN_U8_HWC, _lib.current_stream()))
        else:
            _lib.check(lib.y5_focus_conv_u8(C.byref(d), _lib.current_stream()))
        outs.append(y)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and float(outs[0].abs().max()) > 0
    assert lib.y5_focus_conv_u8_hwc(C.byref(d), _lib.IN_U8, _lib.current_stream()) == _lib.EINVAL


@pytest.mark.parametrize('shape', [(1, 32, 64), (2, 64, 128), (3, 96, 192), (2, 640, 640)], ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('mode', ['bias', 'bn'])
def test_focus_and_the_stride2_conv_in_one_kernel(dev, shape, mode):
    """y5_focus_conv2_f16 (yolov5s layers 0 + 1 as one kernel, the tensor between them never written) against
    y5_focus_conv_f16 followed by y5_conv_f16: for bias-only layers bit-identical to the two kernels whose arithmetic order
    it keeps (the Focus kernel; the persistent stride-2 variant); within the convolution tolerance of the rule-chosen
    second kernel."""
    lib = _lib.lib()
    B, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    st = _lib.current_stream()
    img = torch.randn(B, 3, H, W, generator=g).half().to(dev)
    w0p = torch.zeros(128, 192, dtype=torch.float16, device=dev)
    wv = torch.zeros(32, 3, 3, 16, dtype=torch.float16, device=dev)
    wv[..., :12] = (torch.randn(32, 3, 3, 12, generator=g) / 108 ** 0.5).half().to(dev)
    w0p[:32, :144] = wv.reshape(32, 144)
    w1p = torch.zeros(256, 320, dtype=torch.float16, device=dev)
    w1p[:64, :288] = (torch.randn(64, 288, generator=g) / 288 ** 0.5).half().to(dev)
    p0 = [torch.randn(32, generator=g).to(dev), (torch.rand(32, generator=g) + 0.5).to(dev), torch.randn(32, generator=g).to(dev)]
    p1 = [torch.randn(64, generator=g).to(dev), (torch.rand(64, generator=g) + 0.5).to(dev), torch.randn(64, generator=g).to(dev)]
    mid = torch.zeros(B, H // 2, W // 2, 32, dtype=torch.float16, device=dev)
    d0, d1 = _lib.ConvDesc(), _lib.ConvDesc()
    d0.x, d0.w, d0.y = img.data_ptr(), w0p.data_ptr(), mid.data_ptr()
    d0.batch, d0.h, d0.w_in, d0.cin = B, H, W, 3
    d0.ho, d0.wo, d0.cout, d0.y_ld = H // 2, W // 2, 32, 32
    d0.ksize, d0.stride, d0.pad, d0.kpad, d0.w_rows, d0.act, d0.slope = 3, 1, 1, 192, 128, _lib.ACT_LEAKY, 0.1
    d1.x, d1.w = mid.data_ptr(), w1p.data_ptr()
    d1.batch, d1.h, d1.w_in, d1.cin, d1.x_ld = B, H // 2, W // 2, 32, 32
    d1.ho, d1.wo, d1.cout, d1.y_ld = H // 4, W // 4, 64, 64
    d1.ksize, d1.stride, d1.pad, d1.kpad, d1.w_rows, d1.act, d1.slope = 3, 2, 1, 320, 256, _lib.ACT_LEAKY, 0.1
    for d, p in ((d0, p0), (d1, p1)):
        if mode == 'bias':
            d.bias = p[0].data_ptr()
        else:
            d.scale, d.shift = p[1].data_ptr(), p[2].data_ptr()
    outs = {}
    for name, variant in (('rule', 0), ('persistent', lib.y5_conv_num_variants() - 1)):
        y = torch.full((B, H // 4, W // 4, 64), float('nan'), dtype=torch.float16, device=dev)
        d1.y, d1.variant = y.data_ptr(), variant
        _lib.check(lib.y5_focus_conv_f16(C.byref(d0), st))
        _lib.check(lib.y5_conv_f16(C.byref(d1), st))
        outs[name] = y
    ys = []
    for v in (0, 1):  # tile by tile / layer 1 of a tile beside layer 0 of the next: the same sums in the same order
        y = torch.full((B, H // 4, W // 4, 64), float('nan'), dtype=torch.float16, device=dev)
        d1.y, d1.variant, d0.y, d0.variant = y.data_ptr(), 0, None, v
        _lib.check(lib.y5_focus_conv2_f16(C.byref(d0), C.byref(d1), st))
        torch.cuda.synchronize()
        assert not bool(torch.isnan(y).any())
        if mode == 'bias':
            assert torch.equal(y, outs['persistent'])
        else:  # the scale + shift step is one fused multiply-add here: isolated values one fp16 step from the two-kernel path
            close(y.permute(0, 3, 1, 2), outs['persistent'].permute(0, 3, 1, 2), rel=1e-3)
            assert float((y == outs['persistent']).float().mean()) > 0.999
        close(y.permute(0, 3, 1, 2), outs['rule'].permute(0, 3, 1, 2))
        ys.append(y)
    assert torch.equal(ys[0], ys[1])
    d0.variant = 2
    assert lib.y5_focus_conv2_f16(C.byref(d0), C.byref(d1), st) == _lib.EINVAL
    d0.variant = 0
    d1.ho = d1.ho + 1
    assert lib.y5_focus_conv2_f16(C.byref(d0), C.byref(d1), st) == _lib.EINVAL


def test_conv_into_channel_slice_leaves_neighbours_alone(dev):
    """Concat in place: writing a 32-channel result into channels [64, 96) of a 128-channel buffer."""
    lib = _lib.lib()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 20, 20, 64, generator=g).half().to(dev)
    w = torch.zeros(128, 64, dtype=torch.float16, device=dev)
    w[:32] = (torch.randn(32, 64, generator=g) / 8).half().to(dev)
    buf = torch.full((2, 20, 20, 128), 7.0, dtype=torch.float16, device=dev)
    d = _lib.ConvDesc()
    d.x, d.w, d.y = x.data_ptr(), w.data_ptr(), buf.data_ptr() + 2 * 64
    d.batch, d.h, d.w_in, d.cin, d.x_ld = 2, 20, 20, 64, 64
    d.ho, d.wo, d.cout, d.y_ld = 20, 20, 32, 128
    d.ksize, d.stride, d.pad, d.kpad, d.w_rows = 1, 1, 0, 64, 128
    _lib.check(lib.y5_conv_f16(C.byref(d), _lib.current_stream()))
    torch.cuda.synchronize()
    assert bool((buf[..., :64] == 7).all()) and bool((buf[..., 96:] == 7).all())
    ref = (x.float().view(-1, 64) @ w[:32].float().t()).half().view(2, 20, 20, 32)
    close(buf[..., 64:96], ref)


def test_input_of_two_gib_or_more_is_convolved_in_image_slices(dev):
    """The im2col kernel addresses its input by 32-bit byte offsets; y5_conv_f16 splits a batch whose input spans 2 GiB
    or more into runs of whole images.  Every image of such a batch must equal the same image convolved alone."""
    lib = _lib.lib()
    B, H, W, Ci, Co = 9, 1024, 1024, 128, 8      # 9 x 256 MiB of input: slices of 7 + 2 images
    g = torch.Generator().manual_seed(3)
    x = torch.empty(B, H, W, Ci, dtype=torch.float16, device=dev)
    x[0] = torch.randn(H, W, Ci, generator=g).half().to(dev)
    for b in range(1, B):                         # image b = image 0 times (b + 1): distinct images, cheap to make
        x[b] = x[0] * (b + 1)
    w = (torch.randn(Co, Ci, generator=g) / Ci ** 0.5).half().to(dev)
    wp = torch.zeros(256, 128, dtype=torch.float16, device=dev)
    wp[:Co] = w

    def conv(xb):
        y = torch.full((xb.shape[0], H, W, Co), float('nan'), dtype=torch.float16, device=dev)
        d = _lib.ConvDesc()
        d.x, d.w, d.y = xb.data_ptr(), wp.data_ptr(), y.data_ptr()
        d.batch, d.h, d.w_in, d.cin, d.x_ld = xb.shape[0], H, W, Ci, Ci
        d.ho, d.wo, d.cout, d.y_ld = H, W, Co, Co
        d.ksize, d.stride, d.pad, d.kpad, d.w_rows = 1, 1, 0, 128, 256
        d.act, d.slope = _lib.ACT_NONE, 0.0
        _lib.check(lib.y5_conv_f16(C.byref(d), _lib.current_stream()))
        torch.cuda.synchronize()
        return y

    assert x.numel() * 2 >= 2 ** 31
    whole = conv(x)
    assert not bool(torch.isnan(whole).any())
    for b in (0, 6, 7, 8):                        # both sides of the slice boundary
        assert torch.equal(whole[b], conv(x[b:b + 1])[0])
    ref = (x[8, :64, :64].float() @ w.float().t()).half()
    close(whole[8, :64, :64].permute(2, 0, 1)[None], ref.permute(2, 0, 1)[None])


def _desc(dev, x_nhwc, w, bias=None, scale=None, shift=None, slope=None, res=None, stride=1, keep=None):
    """y5_conv_desc for NHWC fp16 x (B,H,W,Cpad) and weights (Co,Ci,k,k); returns (desc, output tensor, objects to keep)."""
    B, H, W, cip = x_nhwc.shape
    Co, Ci, k, _ = w.shape
    cop = (Co + 7) // 8 * 8
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    kk = k * k * cip
    kpad = (kk + 63) // 64 * 64
    wp = torch.zeros(256, kpad, dtype=torch.float16, device=dev)
    wv = torch.zeros(Co, k, k, cip, dtype=torch.float16, device=dev)
    wv[..., :Ci] = w.permute(0, 2, 3, 1)
    wp[:Co, :kk] = wv.reshape(Co, kk)
    y = torch.full((B, Ho, Wo, cop), float('nan'), dtype=torch.float16, device=dev)
    keep = [wp, y]

    def vec(t):
        if t is None:
            return None
        v = torch.zeros(cop, dtype=torch.float32, device=dev)
        v[:Co] = t.float()
        keep.append(v)
        return v.data_ptr()
    d = _lib.ConvDesc()
    d.x, d.w, d.y = x_nhwc.data_ptr(), wp.data_ptr(), y.data_ptr()
    d.bias, d.scale, d.shift = vec(bias), vec(scale), vec(shift)
    if res is not None:
        d.res, d.res_ld = res.data_ptr(), res.shape[-1]
    d.batch, d.h, d.w_in, d.cin, d.x_ld = B, H, W, cip, cip
    d.ho, d.wo, d.cout, d.y_ld = Ho, Wo, cop, cop
    d.ksize, d.stride, d.pad, d.kpad, d.w_rows = k, stride, k // 2, kpad, 256
    d.act, d.slope = (_lib.ACT_LEAKY, slope) if slope is not None else (_lib.ACT_NONE, 0.0)
    return d, y, keep


@pytest.mark.parametrize('shape', [(2, 32, 32, 32, 40, 24, 3, 1), (3, 64, 64, 64, 20, 20, 3, 1), (1, 128, 128, 128, 20, 12, 3, 1),
                                   (2, 24, 40, 56, 17, 9, 3, 1), (2, 64, 64, 32, 24, 24, 1, 1), (1, 64, 128, 64, 16, 16, 3, 2),
                                   (1, 64, 64, 64, 8, 16, 3, 1), (2, 64, 64, 64, 16, 48, 3, 1), (5, 64, 64, 64, 80, 80, 3, 1),
                                   # the weights-in-registers 128-channel kernel: whole and ragged strips, narrow and full-width rows
                                   (32, 128, 128, 128, 40, 40, 3, 1), (2, 128, 128, 128, 33, 40, 3, 1), (3, 128, 128, 128, 7, 5, 3, 1),
                                   (2, 128, 128, 128, 4, 37, 3, 1)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_conv_chain_equals_two_convolutions(dev, shape):
    """y5_conv_chain_f16 (conv + residual, then a 1x1 conv of the result, one kernel) against the same two convolutions
    as separate y5_conv_f16 calls: equal within the conv tolerance for every tile shape that holds both channel counts,
    with and without the intermediate tensor being stored."""
    lib = _lib.lib()
    B, Ci, Cm, Co, H, W, k, s = shape
    g = torch.Generator().manual_seed(sum(shape))
    st = _lib.current_stream()
    cip = (Ci + 7) // 8 * 8
    x = torch.zeros(B, H, W, cip, dtype=torch.float16, device=dev)
    x[..., :Ci] = torch.randn(B, H, W, Ci, generator=g).half().to(dev)
    w1 = (torch.randn(Cm, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5).half().to(dev)
    w2 = (torch.randn(Co, Cm, 1, 1, generator=g) / Cm ** 0.5).half().to(dev)
    b1 = torch.randn(Cm, generator=g).to(dev)
    sc2, sh2 = (torch.rand(Co, generator=g) + 0.5).to(dev), torch.randn(Co, generator=g).to(dev)
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    cmp_ = (Cm + 7) // 8 * 8
    res = torch.zeros(B, Ho, Wo, cmp_, dtype=torch.float16, device=dev)
    res[..., :Cm] = torch.randn(B, Ho, Wo, Cm, generator=g).half().to(dev)
    for use_res in (True, False):
        d1, y1, k1 = _desc(dev, x, w1, bias=b1, slope=0.1, res=res if use_res else None, stride=s)
        _lib.check(lib.y5_conv_f16(C.byref(d1), st))
        d2, y2, k2 = _desc(dev, y1, w2, scale=sc2, shift=sh2, slope=0.1)
        _lib.check(lib.y5_conv_f16(C.byref(d2), st))
        torch.cuda.synchronize()
        y1_ref, y2_ref = y1.clone(), y2.clone()
        taken = 0
        for v in range(lib.y5_conv_chain_num_variants() + 1):
            for keep_first in (1, 0):
                y1.fill_(float('nan'))
                y2.fill_(float('nan'))
                if not keep_first:
                    d1.y = None
                else:
                    d1.y = y1.data_ptr()
                rc = lib.y5_conv_chain_f16(C.byref(d1), C.byref(d2), keep_first, v, st)
                if v == lib.y5_conv_chain_num_variants() and (Ci, Cm, Co, k, s) == (64, 64, 64, 3, 1) and H % 8 == 0 and W % 16 == 0:
                    assert rc == 0, 'the persistent 64-channel kernel takes this shape'
                if v == lib.y5_conv_chain_num_variants() - 2 and (Ci, Cm, Co, k, s) == (128, 128, 128, 3, 1) and W <= 40:
                    assert rc == 0, 'the 128-channel weights-in-registers kernel takes this shape'
                if v and rc == _lib.EINVAL:
                    continue
                _lib.check(rc)
                torch.cuda.synchronize()
                taken += 1
                if keep_first:
                    close(y1.permute(0, 3, 1, 2), y1_ref.permute(0, 3, 1, 2))
                else:
                    assert bool(torch.isnan(y1).all()), 'the intermediate tensor must not be written'
                assert not bool(torch.isnan(y2).any())
                close(y2.permute(0, 3, 1, 2), y2_ref.permute(0, 3, 1, 2), rel=4e-3)
        assert taken >= 4


@pytest.mark.parametrize('shape', [(3, 256, 256, 128, 128, 21, 23), (32, 256, 256, 128, 128, 40, 40), (2, 256, 256, 128, 128, 17, 9),
                                   (2, 128, 128, 64, 64, 33, 31), (8, 128, 128, 64, 64, 80, 80), (1, 128, 128, 64, 64, 1, 5)],
                         ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('mode', ['csp', 'bias'])
def test_persistent_1x1_chain_on_leading_channels(dev, shape, mode):
    """The persistent 1x1 kernel with a second 1x1 convolution of the LEADING channels of its result (BottleneckCSP's
    cv1 | cv2 followed by the first Bottleneck's cv1, models/common.py:59-60) against the same two convolutions as
    separate y5_conv_f16 calls; pixel counts that are not whole tiles, fewer and more tiles than workgroups."""
    lib = _lib.lib()
    B, Ci, Cm, Cs, Co, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    st = _lib.current_stream()
    x = torch.randn(B, H, W, Ci, generator=g).half().to(dev)
    w1 = (torch.randn(Cm, Ci, 1, 1, generator=g) / Ci ** 0.5).half().to(dev)
    w2 = (torch.randn(Co, Cs, 1, 1, generator=g) / Cs ** 0.5).half().to(dev)
    b1, b2 = torch.randn(Cm, generator=g).to(dev), torch.randn(Co, generator=g).to(dev)
    sc1, sh1 = (torch.rand(Cm, generator=g) + 0.5).to(dev), torch.randn(Cm, generator=g).to(dev)
    if mode == 'csp':  # what the engine records: bias + scale + shift on the merged cv1 | cv2, a folded bias on m[0].cv1
        d1, y1, k1 = _desc(dev, x, w1, bias=b1, scale=sc1, shift=sh1, slope=0.1)
    else:
        d1, y1, k1 = _desc(dev, x, w1, bias=b1, slope=0.1)
    _lib.check(lib.y5_conv_f16(C.byref(d1), st))
    d2, y2, k2 = _desc(dev, y1[..., :Cs], w2, bias=b2, slope=0.1) if mode == 'csp' else _desc(dev, y1[..., :Cs], w2, scale=sc1[:Co], shift=sh1[:Co])
    d2.x_ld = Cm
    _lib.check(lib.y5_conv_f16(C.byref(d2), st))
    torch.cuda.synchronize()
    y1_ref, y2_ref = y1.clone(), y2.clone()
    v = lib.y5_conv_chain_num_variants() - 1
    for variant in (v, 0) if Cs < Cm else (v,):
        for keep_first in (1, 0):
            y1.fill_(float('nan'))
            y2.fill_(float('nan'))
            d1.y = y1.data_ptr() if keep_first else None
            _lib.check(lib.y5_conv_chain_f16(C.byref(d1), C.byref(d2), keep_first, variant, st))
            torch.cuda.synchronize()
            if keep_first:
                close(y1.permute(0, 3, 1, 2), y1_ref.permute(0, 3, 1, 2))
                assert float((y1 == y1_ref).float().mean()) > 0.99
            else:
                assert bool(torch.isnan(y1).all()), 'the intermediate tensor must not be written'
            assert not bool(torch.isnan(y2).any())
            close(y2.permute(0, 3, 1, 2), y2_ref.permute(0, 3, 1, 2), rel=4e-3)
    if Cs < Cm:  # the other kernels of the family read all of the first result
        d1.y = y1.data_ptr()
        assert lib.y5_conv_chain_f16(C.byref(d1), C.byref(d2), 1, 1, st) == _lib.EINVAL


@pytest.mark.parametrize('shape', [(2, 64, 96, 32), (3, 41, 74, 16), (1, 130, 66, 56), (2, 8, 200, 13)], ids=lambda s: 'x'.join(map(str, s)))
def test_stem_conv_from_the_nchw_image(dev, shape):
    """y5_stem_conv_f16 / _u8 (first-layer 3x3 conv straight from the NCHW image) against torch's fp32 convolution with
    the epilogue's rounding points, ragged image sizes and channel counts included."""
    lib = _lib.lib()
    B, H, W, Co = shape
    g = torch.Generator().manual_seed(sum(shape))
    cop = (Co + 7) // 8 * 8
    w = (torch.randn(Co, 3, 3, 3, generator=g) / 27 ** 0.5).half().to(dev)
    wp = torch.zeros(256, 128, dtype=torch.float16, device=dev)
    wv = torch.zeros(Co, 3, 3, 8, dtype=torch.float16, device=dev)
    wv[..., :3] = w.permute(0, 2, 3, 1)
    wp[:Co, :72] = wv.reshape(Co, 72)
    scale = torch.zeros(cop, device=dev); shift = torch.zeros(cop, device=dev)
    scale[:Co] = (torch.rand(Co, generator=g) + 0.5).to(dev); shift[:Co] = torch.randn(Co, generator=g).to(dev)
    img8 = torch.randint(0, 256, (B, 3, H, W), generator=g, dtype=torch.int32).to(torch.uint8).to(dev)
    for u8 in (False, True):
        img = img8 if u8 else (torch.randn(B, 3, H, W, generator=g)).half().to(dev)
        xh = (img.half() / 255.0) if u8 else img
        y = torch.full((B, H, W, cop), float('nan'), dtype=torch.float16, device=dev)
        d = _lib.ConvDesc()
        d.x, d.w, d.y = img.data_ptr(), wp.data_ptr(), y.data_ptr()
        d.scale, d.shift = scale.data_ptr(), shift.data_ptr()
        d.batch, d.h, d.w_in, d.cin, d.x_ld = B, H, W, 3, 0
        d.ho, d.wo, d.cout, d.y_ld = H, W, cop, cop
        d.ksize, d.stride, d.pad, d.kpad, d.w_rows = 3, 1, 1, 128, 256
        d.act, d.slope = _lib.ACT_LEAKY, 0.1
        _lib.check((lib.y5_stem_conv_u8 if u8 else lib.y5_stem_conv_f16)(C.byref(d), _lib.current_stream()))
        torch.cuda.synchronize()
        acc = F.conv2d(xh.float(), w.float(), None, 1, 1)
        ref = (acc.half().float() * scale[:Co].view(1, -1, 1, 1) + shift[:Co].view(1, -1, 1, 1)).half()
        ref = F.leaky_relu(ref.float(), 0.1).half()
        assert not bool(torch.isnan(y).any()), 'stem conv left output elements unwritten'
        assert bool((y[..., Co:] == 0).all()), 'padding channels must be zero'
        close(y[..., :Co].permute(0, 3, 1, 2), ref)


@pytest.mark.parametrize('shape', [(2, 64, 64, 40, 24, 16), (1, 128, 72, 255, 40, 40), (3, 256, 256, 256, 8, 12), (2, 64, 8, 16, 6, 10),
                                   (1, 192, 104, 136, 20, 18),
                                   # 128 + 128 -> 128: also the persistent kernel (whole and ragged pixel runs, fewer pixels than one tile per CU)
                                   (4, 128, 128, 128, 40, 40), (2, 128, 128, 128, 12, 22), (1, 128, 128, 128, 6, 10),
                                   (3, 256, 256, 256, 20, 20), (2, 256, 256, 256, 10, 14)], ids=lambda s: 'x'.join(map(str, s)))
def test_conv_cat2_equals_upsample_concat_conv(dev, shape):
    """y5_conv_cat2_f16 (1x1 conv of [nearest-2x-upsampled a | b], neither tensor built) against upsampling, concatenating
    and y5_conv_f16, for every compiled tile shape; with residual, bias, folded BatchNorm and LeakyReLU."""
    lib = _lib.lib()
    B, Ca, Cb, Co, H, W = shape  # H, W: the output (= b's) size
    g = torch.Generator().manual_seed(sum(shape))
    st = _lib.current_stream()
    cbp, cop = (Cb + 7) // 8 * 8, (Co + 7) // 8 * 8
    a_ld = Ca + 16   # pitched sources
    a = torch.zeros(B, H // 2, W // 2, a_ld, dtype=torch.float16, device=dev)
    a[..., :Ca] = torch.randn(B, H // 2, W // 2, Ca, generator=g).half().to(dev)
    b = torch.zeros(B, H, W, cbp, dtype=torch.float16, device=dev)
    b[..., :Cb] = torch.randn(B, H, W, Cb, generator=g).half().to(dev)
    cat = torch.cat((a[..., :Ca].repeat_interleave(2, 1).repeat_interleave(2, 2), b), 3).contiguous()
    w = torch.zeros(Co, Ca + cbp, 1, 1)
    w[:, :Ca + Cb] = torch.randn(Co, Ca + Cb, 1, 1, generator=g) / (Ca + Cb) ** 0.5
    w = w.half().to(dev)
    sc, sh = (torch.rand(Co, generator=g) + 0.5).to(dev), torch.randn(Co, generator=g).to(dev)
    res = torch.zeros(B, H, W, cop, dtype=torch.float16, device=dev)
    res[..., :Co] = torch.randn(B, H, W, Co, generator=g).half().to(dev)
    for use_res in (False, True):
        d, y, keep = _desc(dev, cat, w, scale=sc, shift=sh, slope=0.1, res=res if use_res else None)
        _lib.check(lib.y5_conv_f16(C.byref(d), st))
        torch.cuda.synchronize()
        ref = y.clone()
        d.x, d.cin, d.x_ld = a.data_ptr(), Ca, a_ld
        taken = 0
        for v in range(lib.y5_conv_cat2_num_variants() + 1):
            y.fill_(float('nan'))
            d.variant = v
            rc = lib.y5_conv_cat2_f16(C.byref(d), b.data_ptr(), cbp, cbp, st)
            if v and rc == _lib.EINVAL:
                continue
            _lib.check(rc)
            torch.cuda.synchronize()
            taken += 1
            close(y.permute(0, 3, 1, 2), ref.permute(0, 3, 1, 2))
        assert taken >= 5
        if (Ca, Cb, Co) in ((128, 128, 128), (256, 256, 256)) and not use_res:
            assert rc == 0, 'the persistent kernel (last variant) must take 128 + 128 -> 128 and 256 + 256 -> 256 without a residual'
        else:
            assert rc == _lib.EINVAL
    # rejected: odd output size, a first source that is not a multiple of 64 channels wide, a 3x3 kernel
    d.variant = 0
    d.cin = 56
    assert lib.y5_conv_cat2_f16(C.byref(d), b.data_ptr(), cbp, cbp, st) == _lib.EINVAL
    d.cin, d.ksize, d.pad = Ca, 3, 1
    assert lib.y5_conv_cat2_f16(C.byref(d), b.data_ptr(), cbp, cbp, st) == _lib.EINVAL
    d.ksize, d.pad, d.ho, d.h = 1, 0, H + 1, H + 1
    assert lib.y5_conv_cat2_f16(C.byref(d), b.data_ptr(), cbp, cbp, st) == _lib.EINVAL
