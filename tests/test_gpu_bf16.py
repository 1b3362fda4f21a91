"""The bfloat16 path (models under .bfloat16()): im2col MFMA convolutions, pooling / upsampling, Detect decode, whole
graphs - against the oracle's precision='bf16' restatement, which tests/test_oracle_golden.py pins to the reference's
modules under .bfloat16(), and against the reference's own bf16 fixtures.

Tolerance: bfloat16 keeps 8 significant bits, so one rounding step is 2^-8 relative (8x an fp16 step); kernels and
oracle add the same products in different fp32 orders, and a value next to a rounding boundary may come out one step
apart: single layers |d| <= 2 ulp_bf16(max(|ref|, largest intermediate)); blocks and graphs, where a first-layer step
is carried through later layers, |d| <= 2^-6 |ref| + 2^-6 max|ref| with most values bit-identical."""
import ctypes as C
import os

import pytest
import torch
import torch.nn.functional as F
import yaml

import util
from oracle import y5ref
from yolo_amd import _lib, engine

pytestmark = pytest.mark.gpu
OPS = y5ref._Ops('bf16')


def close(got, ref, frac, rel=2.0 ** -6):
    got, ref = got.float().cpu(), ref.float().cpu()
    d = (got - ref).abs()
    tol = rel * ref.abs() + rel * ref.abs().max()
    assert bool((d <= tol).all()), 'worst %.2f of the tolerance' % float((d / tol).max())
    exact = float((d == 0).float().mean())
    assert exact >= frac, 'only %.3f bit-identical' % exact
    return exact


@pytest.mark.parametrize('name', ['conv_k3s1', 'csp', 'spp', 'focus'])
def test_block_against_reference_bf16_fixture(dev, name):
    """The HIP block on bf16 tensors against what the reference's module computed under .bfloat16() on the CPU."""
    z = util.golden('bf16block_' + name)
    m, _, _ = util.block_from_fixture(name)
    m.load_state_dict(util.sd_of(z, 'model.0.'), strict=True)
    x, y = torch.from_numpy(z['x']), torch.from_numpy(z['y'])
    got = engine.run_block(m.bfloat16().to(dev), x.bfloat16().to(dev))
    assert got.dtype is torch.bfloat16 and got.shape == y.shape
    assert close(got, y, 0.97) > 0.97


@pytest.mark.parametrize('name', util.BLOCKS)
def test_block_against_oracle(dev, name):
    m, x, _ = util.block_from_fixture(name)
    m = m.bfloat16()
    sd = {'model.0.' + k: v.float() for k, v in m.state_dict().items()}
    xb = x.bfloat16()
    if name.startswith('conv'):
        ref = y5ref.conv_block(OPS, sd, 'model.0', xb.float(), 2 if 's2' in name else 1)
    elif name == 'focus':
        ref = y5ref.conv_block(OPS, sd, 'model.0.conv', y5ref.focus_gather(xb.float()))
    elif name.startswith('bottleneck'):
        ref = y5ref.bottleneck(OPS, sd, 'model.0', xb.float(), 'noadd' not in name)
    elif name.startswith('csp'):
        ref = y5ref.bottleneck_csp(OPS, sd, 'model.0', xb.float(), len(m.m), 'noshortcut' not in name)
    else:
        ref = y5ref.spp(OPS, sd, 'model.0', xb.float())
    got = engine.run_block(m.to(dev), xb.to(dev))
    close(got, ref, 0.97)


def _conv(dev, x, w, bias, slope, stride, variant, res=None):
    """y5_conv_f16 with dtype = Y5_BF16 on NCHW bf16 tensors; None when the variant does not take the shape."""
    lib = _lib.lib()
    B, Ci, H, W = x.shape
    Co, _, k, _ = w.shape
    cip, cop = (Ci + 7) // 8 * 8, (Co + 7) // 8 * 8
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    xn = torch.zeros(B, H, W, cip, dtype=torch.bfloat16, device=dev)
    xn[..., :Ci] = x.permute(0, 2, 3, 1)
    kk = k * k * cip
    kpad, rows = (kk + 63) // 64 * 64, (Co + 255) // 256 * 256
    wp = torch.zeros(rows, kpad, dtype=torch.bfloat16, device=dev)
    wv = torch.zeros(Co, k, k, cip, dtype=torch.bfloat16, device=dev)
    wv[..., :Ci] = w.permute(0, 2, 3, 1)
    wp[:Co, :kk] = wv.reshape(Co, kk)
    y = torch.full((B, Ho, Wo, cop), float('nan'), dtype=torch.bfloat16, device=dev)
    bv = torch.zeros(cop, dtype=torch.float32, device=dev)
    bv[:Co] = bias
    d = _lib.ConvDesc()
    d.dtype = _lib.BF16
    d.x, d.w, d.y, d.bias = xn.data_ptr(), wp.data_ptr(), y.data_ptr(), bv.data_ptr()
    if res is not None:
        rn = torch.zeros(B, Ho, Wo, cop, dtype=torch.bfloat16, device=dev)
        rn[..., :Co] = res.permute(0, 2, 3, 1)
        d.res, d.res_ld = rn.data_ptr(), cop
    d.batch, d.h, d.w_in, d.cin, d.x_ld = B, H, W, cip, cip
    d.ho, d.wo, d.cout, d.y_ld = Ho, Wo, cop, cop
    d.ksize, d.stride, d.pad, d.kpad, d.w_rows = k, stride, k // 2, kpad, rows
    d.act, d.slope, d.variant = _lib.ACT_LEAKY, slope, variant
    rc = lib.y5_conv_f16(C.byref(d), _lib.current_stream())
    if variant and rc == _lib.EINVAL:
        return None
    _lib.check(rc)
    torch.cuda.synchronize()
    assert not bool(torch.isnan(y.float()).any()) and bool((y[..., Co:] == 0).all())
    return y[..., :Co].permute(0, 3, 1, 2)


def _ulp(t):
    a = t.abs().clamp_min(2.0 ** -126)
    return torch.pow(2.0, torch.floor(torch.log2(a)) - 7)


@pytest.mark.parametrize('shape', [(2, 64, 128, 40, 40, 1, 1), (3, 32, 32, 48, 41, 1, 1), (2, 32, 64, 24, 41, 3, 1), (2, 24, 48, 30, 30, 3, 2),
                                   (1, 256, 256, 20, 20, 3, 1), (3, 13, 25, 9, 7, 3, 1), (5, 128, 255, 20, 12, 1, 1)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_every_bf16_conv_variant_against_torch_fp32(dev, shape):
    """Every tile configuration compiled for bf16 (and the rule-based choice) against an fp32 torch convolution of the same
    bf16 operands, with and without the residual."""
    lib = _lib.lib()
    B, Ci, Co, H, W, k, s = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, Ci, H, W, generator=g).bfloat16().to(dev)
    w = (torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5).bfloat16().to(dev)
    bias = torch.randn(Co, generator=g).to(dev)
    acc = F.conv2d(x.float(), w.float(), None, s, k // 2)
    order = F.conv2d(x.float().abs(), w.float().abs(), None, s, k // 2) * 2.0 ** -18
    pre = (acc + bias.view(1, -1, 1, 1)).bfloat16().float()
    ref = F.leaky_relu(pre, 0.1).bfloat16()
    res = torch.randn(ref.shape, generator=g).bfloat16().to(dev)
    ref_res = (ref.float() + res.float()).bfloat16()
    taken = 0
    assert lib.y5_conv_num_variants_bf16() >= 8
    for v in range(lib.y5_conv_num_variants_bf16() + 1):
        got = _conv(dev, x, w, bias, 0.1, s, v)
        if got is None:
            continue
        taken += 1
        for out, want, mags in ((got, ref, (acc, bias.view(1, -1, 1, 1).expand_as(acc))),
                                (_conv(dev, x, w, bias, 0.1, s, v, res=res), ref_res, (acc, bias.view(1, -1, 1, 1).expand_as(acc), ref.float(), res.float()))):
            m = want.float().abs()
            for t in mags:
                m = torch.maximum(m, t.abs())
            tol = 2.0 * _ulp(m) + order
            dd = (out.float() - want.float()).abs()
            assert bool((dd <= tol).all()), 'variant %d: worst %.2f of the tolerance' % (v, float((dd / tol).max()))
    assert taken >= 4
    d = _lib.ConvDesc()
    d.dtype = 7
    assert lib.y5_conv_f16(C.byref(d), _lib.current_stream()) == _lib.EINVAL


def test_decode_bit_exact_against_reference_bf16_fixture(dev):
    """y5_detect_decode_multi_f16 with dtype = Y5_BF16 against Detect.forward under .bfloat16() (reference fixture)."""
    lib = _lib.lib()
    z = util.golden('decode_bf16_64x96')
    zg = torch.from_numpy(z['z'])
    no = zg.shape[2]
    zo = torch.zeros(zg.shape, dtype=torch.bfloat16, device=dev)
    levels = (_lib.DecodeDesc * 3)()
    keep, xs, row0 = [], [], 0
    for i in range(3):
        raw = torch.from_numpy(z['raw%d' % i]).bfloat16().to(dev)
        B, c, ny, nx = raw.shape
        na, ld = c // no, (c + 7) // 8 * 8
        head = torch.zeros(B, ny, nx, ld, dtype=torch.bfloat16, device=dev)
        head[..., :c] = raw.permute(0, 2, 3, 1)
        xo = torch.zeros(B, na, ny, nx, no, dtype=torch.bfloat16, device=dev)
        keep.append(head)
        xs.append(xo)
        d = levels[i]
        d.dtype = _lib.BF16
        d.head, d.head_ld, d.x_out, d.z_out = head.data_ptr(), ld, xo.data_ptr(), zo.data_ptr()
        d.batch, d.ny, d.nx, d.na, d.no = B, ny, nx, na, no
        d.z_rows_total, d.z_row0, d.stride = zg.shape[1], row0, float(z['stride'][i])
        wh = z['anchor_grid'][i].reshape(-1, 2)
        for a in range(na):
            d.anchor_wh[2 * a], d.anchor_wh[2 * a + 1] = float(wh[a, 0]), float(wh[a, 1])
        row0 += na * ny * nx
    _lib.check(lib.y5_detect_decode_multi_f16(levels, 3, _lib.current_stream()))
    torch.cuda.synchronize()
    for i in range(3):
        assert torch.equal(xs[i].float().cpu(), torch.from_numpy(z['x%d' % i]))
    assert torch.equal(zo.float().cpu(), zg)


def test_pooling_and_bilinear_upsampling_in_bf16(dev):
    lib = _lib.lib()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 16, 12, 10, generator=g).bfloat16().to(dev)
    xn = x.permute(0, 2, 3, 1).contiguous()
    cat = torch.zeros(2, 12, 10, 64, dtype=torch.bfloat16, device=dev)
    cat[..., :16] = xn
    _lib.check(lib.y5_maxpool_cascade_x16(cat.data_ptr(), 64, cat.data_ptr() + 32, 64, 16, 3, 2, 12, 10, 16, 5, _lib.BF16, _lib.current_stream()))
    ref = torch.cat([x] + [F.max_pool2d(x.float(), k, 1, k // 2).bfloat16() for k in (5, 9, 13)], 1)
    assert torch.equal(cat.permute(0, 3, 1, 2), ref)
    y = torch.zeros(2, 12, 10, 16, dtype=torch.bfloat16, device=dev)
    _lib.check(lib.y5_maxpool_x16(xn.data_ptr(), 16, y.data_ptr(), 16, 2, 12, 10, 16, 9, _lib.BF16, _lib.current_stream()))
    assert torch.equal(y.permute(0, 3, 1, 2), F.max_pool2d(x.float(), 9, 1, 4).bfloat16())
    up = torch.zeros(2, 24, 20, 16, dtype=torch.bfloat16, device=dev)
    _lib.check(lib.y5_upsample2x_x16(xn.data_ptr(), 16, up.data_ptr(), 16, 2, 12, 10, 16, 1, _lib.BF16, _lib.current_stream()))
    want = F.interpolate(x.float().cpu(), scale_factor=2.0, mode='bilinear', align_corners=False).bfloat16()
    d = (up.permute(0, 3, 1, 2).float().cpu() - want.float()).abs()
    assert bool((d <= _ulp(want.float())).all())   # fp32 blend rounded once: at most one bf16 step from torch's kernel


@pytest.mark.parametrize('cfg,fuse,shape', [('yolov5s', True, (2, 3, 64, 96)), ('yolov5s', False, (1, 3, 128, 64)),
                                            ('yolov5s_ori', True, (2, 3, 96, 64))])
def test_bf16_forward_against_oracle(dev, cfg, fuse, shape):
    """Model.forward under .bfloat16() on the engine (im2col convolutions, stand-alone decode) against the oracle."""
    model = util.build_model(cfg)
    util.randomize_bn(model, 11)
    model.eval()
    if fuse:
        util.quiet(model.fuse)
    model = model.bfloat16().to(dev)
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(5)).bfloat16()
    z, xs = model(x.to(dev))
    assert engine.calls(model) == 1 and z.dtype is torch.bfloat16
    cfgd = yaml.safe_load(open(os.path.join(util.CFG_DIR, cfg + '.yaml')))
    sd = {k: v.float().cpu() for k, v in model.state_dict().items()}
    zr, xr = y5ref.forward(cfgd, sd, x.float(), precision='bf16', strides=model.stride)
    for a, b in zip(xs, xr):
        close(a, b, 0.5, rel=2.0 ** -5)
    close(z, zr, 0.8, rel=2.0 ** -5)
    with engine.disabled():
        zt, _ = model(x.to(dev))
    close(z, zt, 0.5, rel=2.0 ** -5)
