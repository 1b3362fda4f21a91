"""GPU parity of the BatchNorm-recalibration forward (Training.adaptive_BN, filter_pruning/fp_train.py:207-261).

The reference runs this forward in fp32; the HIP path keeps activations and weights in fp16 (sums and statistics in
fp32/fp64), so the recalibrated statistics agree to fp16 accuracy.  Tolerances, stated where used:
  * kernels on their own: against fp64 torch arithmetic on the same fp16 inputs, 2e-6 relative for the sums,
    one fp16 rounding for the normalised output;
  * a few blocks deep (Conv -> BottleneckCSP -> SPP -> Bottleneck -> Conv, 13 BatchNorms), statistics of ONE batch
    (momentum 1.0): running_mean within 0.002 * sqrt(running_var) + 5e-4, running_var within 0.4 % + 1e-4 of the fp32
    modules (measured: 0.04 % of a standard deviation, 0.02 %);
  * whole model, statistics of ONE batch: running_mean within 0.03 * sqrt(running_var) + 2e-3, running_var within
    6 % + 1e-4 of the fp32 oracle, at sizes that leave >= 500 values per channel in the deepest layer.  A train-mode
    stack of ~60 BatchNorms amplifies rounding-level differences (the oracle restated with fp16 activations differs
    from the fp32 oracle as much as the kernels do: 0.7 % of a standard deviation / 2 % of a variance at
    16 x 320 x 320, several times that with under 100 values per channel or on the deeper yolov5m), which is why the tight comparison is
    made on the shallow stack above;
  * whole model against the reference's own run (golden fixture, momentum 0.03, two batches, so a batch's statistics
    enter with weight 0.03): running_mean within 2e-3 * sqrt(running_var) + 2e-4, running_var within 0.4 % + 1e-5.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F
import yaml

from tests import util

pytestmark = pytest.mark.gpu


def _lib():
    from yolo_amd import _lib as L
    return L, L.lib()


@pytest.mark.parametrize('pixels,c,ld', [(1000, 8, 8), (4096, 64, 64), (777, 104, 128), (20000, 32, 96), (300000, 16, 16),
                                         (50, 1280, 1280), (3, 2048, 2048), (3276800, 32, 32)])
def test_bn_stats_sums(pixels, c, ld):
    L, lib = _lib()
    g = torch.Generator().manual_seed(pixels + c)
    x = (torch.randn(pixels, ld, generator=g) * 2 + 0.5).half().cuda()
    rows = lib.y5_bn_stats_rows(pixels, c)
    assert 1 <= rows <= 512
    part = torch.full((rows + 1, 2 * c), 7.0, dtype=torch.float32, device='cuda')   # overwritten, not accumulated into
    L.check(lib.y5_bn_stats_f16(x.data_ptr(), ld, pixels, c, part.data_ptr(), L.current_stream()))
    again = torch.zeros_like(part)
    L.check(lib.y5_bn_stats_f16(x.data_ptr(), ld, pixels, c, again.data_ptr(), L.current_stream()))
    torch.cuda.synchronize()
    assert bool((part[rows] == 7.0).all())            # exactly `rows` rows are written
    assert torch.equal(part[:rows], again[:rows])     # no atomics on the way: bit-identical on repetition
    sums = part[:rows].double().sum(0)
    xd = x[:, :c].double()
    ref = torch.cat((xd.sum(0), (xd * xd).sum(0)))
    # fp32 partial sums of at most a few thousand values, fp64 above: 2e-6 of the sum of magnitudes
    bound = 2e-6 * torch.cat((xd.abs().sum(0), (xd * xd).sum(0))) + 1e-9
    assert bool(((sums - ref).abs() <= bound).all())


@pytest.mark.parametrize('c,c_valid,count,momentum,rows', [(64, 64, 4096, 0.03, 1), (16, 13, 100, 1.0, 5), (8, 5, 1, 0.5, 1),
                                                           (104, 100, 3000, 0.1, 37)])
def test_bn_finalize_matches_batch_norm(c, c_valid, count, momentum, rows):
    L, lib = _lib()
    g = torch.Generator().manual_seed(c)
    x = (torch.randn(count, c_valid, generator=g) * 3 + 1).half().double()
    part = torch.zeros(rows, 2 * c, dtype=torch.float32)
    for r, chunk in enumerate(torch.chunk(x, rows, 0)):   # the rows are partial sums of disjoint shares
        part[r, :c_valid], part[r, c:c + c_valid] = chunk.sum(0).float(), (chunk * chunk).sum(0).float()
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    rm, rv = torch.randn(c, generator=g), torch.rand(c, generator=g) + 0.5
    rm_ref, rv_ref = rm[:c_valid].clone().double(), rv[:c_valid].clone().double()
    if count > 1:
        y_ref = F.batch_norm(x.t()[None], rm_ref, rv_ref, gamma[:c_valid].double(), beta[:c_valid].double(), True, momentum, 1e-3)[0].t()
    else:  # torch refuses one value per channel in train mode; the formulas still hold
        y_ref = beta[:c_valid].double().expand(1, -1)
        rm_ref = (1 - momentum) * rm_ref + momentum * x[0]
        rv_ref = (1 - momentum) * rv_ref
    d = lambda t: t.cuda()
    part_d, gamma_d, beta_d, rm_d, rv_d = d(part), d(gamma), d(beta), d(rm), d(rv)
    scale, shift = torch.full((c,), 9.0, device='cuda'), torch.full((c,), 9.0, device='cuda')
    L.check(lib.y5_bn_finalize_f32(part_d.data_ptr(), rows, count, c, c_valid, gamma_d.data_ptr(), beta_d.data_ptr(), 1e-3, momentum,
                                   rm_d.data_ptr(), rv_d.data_ptr(), scale.data_ptr(), shift.data_ptr(), L.current_stream()))
    torch.cuda.synchronize()
    # the partial sums handed in are fp32: 1e-5 covers their rounding
    assert torch.allclose(rm_d[:c_valid].cpu().double(), rm_ref, rtol=1e-5, atol=1e-5)
    assert torch.allclose(rv_d[:c_valid].cpu().double(), rv_ref, rtol=1e-5, atol=1e-5)
    y = x * scale[:c_valid].cpu().double() + shift[:c_valid].cpu().double()
    assert torch.allclose(y, y_ref, rtol=1e-4, atol=1e-4)   # scale / shift are fp32 vectors
    assert bool((scale[c_valid:] == 0).all()) and bool((shift[c_valid:] == 0).all())
    assert torch.equal(rm_d[c_valid:].cpu(), rm[c_valid:]) and torch.equal(rv_d[c_valid:].cpu(), rv[c_valid:])


@pytest.mark.parametrize('pixels,c,ld,res,act', [(1000, 8, 8, False, 1), (513, 104, 128, True, 1), (4096, 64, 64, True, 0),
                                                 (77, 1280, 1280, False, 0)])
def test_affine_act(pixels, c, ld, res, act):
    L, lib = _lib()
    g = torch.Generator().manual_seed(pixels)
    x = torch.randn(pixels, ld, generator=g).half().cuda()
    r = torch.randn(pixels, ld, generator=g).half().cuda()
    scale, shift = (torch.rand(c, generator=g) + 0.5).cuda(), torch.randn(c, generator=g).cuda()
    y = torch.full_like(x, 3.0)
    L.check(lib.y5_affine_act_f16(x.data_ptr(), ld, y.data_ptr(), ld, r.data_ptr() if res else None, ld, pixels, c,
                                  scale.data_ptr(), shift.data_ptr(), act, 0.1, L.current_stream()))
    torch.cuda.synchronize()
    ref = x[:, :c].double() * scale.double() + shift.double()
    if act:
        ref = torch.where(ref > 0, ref, ref * 0.1)
    if res:
        ref = ref + r[:, :c].double()
    # evaluated in fp32, rounded once to fp16
    assert torch.allclose(y[:, :c].double(), ref, rtol=1.5e-3, atol=1e-4)
    assert bool((y[:, c:] == 3.0).all())   # the pitch gap is not written
    # in place
    L.check(lib.y5_affine_act_f16(x.data_ptr(), ld, x.data_ptr(), ld, r.data_ptr() if res else None, ld, pixels, c,
                                  scale.data_ptr(), shift.data_ptr(), act, 0.1, L.current_stream()))
    torch.cuda.synchronize()
    assert torch.equal(x[:, :c], y[:, :c])


def test_bn_entry_points_reject_bad_arguments():
    L, lib = _lib()
    x = torch.zeros(64, 16, dtype=torch.float16, device='cuda')
    sums = torch.zeros(512 * 32, dtype=torch.float32, device='cuda')
    assert lib.y5_bn_stats_rows(64, 12) == 0 and lib.y5_bn_stats_rows(0, 16) == 0 and lib.y5_bn_stats_rows(64, 4096) == 0
    f = torch.zeros(16, device='cuda')
    s = L.current_stream()
    assert lib.y5_bn_stats_f16(x.data_ptr(), 16, 64, 12, sums.data_ptr(), s) != 0        # channels not a multiple of 8
    assert lib.y5_bn_stats_f16(x.data_ptr(), 8, 64, 16, sums.data_ptr(), s) != 0         # pitch below channels
    assert lib.y5_bn_stats_f16(x.data_ptr(), 16, 0, 16, sums.data_ptr(), s) != 0         # empty
    assert lib.y5_bn_stats_f16(None, 16, 64, 16, sums.data_ptr(), s) != 0
    assert lib.y5_bn_finalize_f32(sums.data_ptr(), 1, 0, 16, 16, None, None, 1e-3, 0.03, None, None, f.data_ptr(), f.data_ptr(), s) != 0
    assert lib.y5_bn_finalize_f32(sums.data_ptr(), 1, 10, 16, 17, None, None, 1e-3, 0.03, None, None, f.data_ptr(), f.data_ptr(), s) != 0
    assert lib.y5_bn_finalize_f32(sums.data_ptr(), 1, 10, 16, 16, None, None, 1e-3, 1.5, None, None, f.data_ptr(), f.data_ptr(), s) != 0
    assert lib.y5_bn_finalize_f32(sums.data_ptr(), 0, 10, 16, 16, None, None, 1e-3, 0.03, None, None, f.data_ptr(), f.data_ptr(), s) != 0
    assert lib.y5_affine_act_f16(x.data_ptr(), 16, x.data_ptr(), 16, None, 0, 64, 16, f.data_ptr(), f.data_ptr(), 7, 0.1, s) != 0
    assert lib.y5_affine_act_f16(x.data_ptr(), 16, x.data_ptr(), 16, None, 0, 64, 16, None, f.data_ptr(), 0, 0.1, s) != 0


def _bn_keys(sd):
    return [k for k in sd if k.endswith('running_mean') or k.endswith('running_var')]


@pytest.mark.parametrize('cfg', ['yolov5s', 'yolov5s_ori'])
def test_recalibrated_statistics_match_the_reference_run(cfg):
    """Two uint8 batches at the models' own momentum, against what the reference's modules produced."""
    from yolo_amd.adaptive_bn import adaptive_bn
    z = util.golden('adaptive_bn_' + cfg)
    model = util.build_model(cfg)
    util.randomize_bn(model, 11)
    tracked0 = int(next(m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)).num_batches_tracked)
    model = model.cuda()
    out = adaptive_bn(model, [(torch.from_numpy(z['x0']), None, None, None), (torch.from_numpy(z['x1']), None, None, None)], deepcp=False)
    assert out is model and model.training
    sd = model.state_dict()
    worst = 0.0
    for k in _bn_keys(sd):
        got, want = sd[k].cpu().double(), torch.from_numpy(z[k]).double()
        if k.endswith('running_mean'):
            tol = 2e-3 * torch.from_numpy(z[k.replace('running_mean', 'running_var')]).double().sqrt() + 2e-4
        else:
            tol = 4e-3 * want + 1e-5
        worst = max(worst, ((got - want).abs() / tol).max().item())
    print('worst error %.3f of the tolerance' % worst)
    assert worst <= 1.0
    assert int(next(m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)).num_batches_tracked) == tracked0 + 2


@pytest.mark.parametrize('cfg,pruned,shape', [('yolov5s', False, (16, 3, 320, 320)), ('yolov5s_ori', True, (8, 3, 256, 256)),
                                              ('yolov5s', False, (8, 3, 256, 384))])
def test_batch_statistics_match_the_oracle(cfg, pruned, shape):
    """Momentum 1.0: the running statistics afterwards ARE the last batch's statistics at every depth of the graph."""
    from oracle import y5ref
    from yolo_amd.adaptive_bn import Recalibrator
    model = util.build_model(cfg)
    cfg_file = cfg
    if pruned:
        util.prune_like(model)
    util.randomize_bn(model, 5)
    bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    for m in bns:
        m.momentum = 1.0
    g = torch.Generator().manual_seed(1)
    batches = [torch.randint(0, 256, shape, generator=g, dtype=torch.uint8) for _ in range(2)]
    if pruned:   # the oracle walks (yaml, state_dict); a pruned model's widths are not in the yaml, so use the modules
        ref_model = __import__('copy').deepcopy(model).train()
        with torch.no_grad():
            for b in batches:
                ref_model(b.float() / 255.0)
        ref = ref_model.state_dict()
    else:
        cfgd = yaml.safe_load(open(os.path.join(util.CFG_DIR, cfg_file + '.yaml')))
        ref = y5ref.adaptive_bn(cfgd, model.state_dict(), batches, momentum=1.0)
    model = model.cuda().train()
    rec = Recalibrator(model)
    for b in batches:
        rec.step(b.cuda())
    rec.finish()
    sd = model.state_dict()
    worst_m = worst_v = 0.0
    for k in _bn_keys(sd):
        got, want = sd[k].cpu().double(), ref[k].double()
        if k.endswith('running_mean'):
            std = ref[k.replace('running_mean', 'running_var')].double().sqrt()
            err = ((got - want).abs() / (0.03 * std + 2e-3)).max().item()
            worst_m = max(worst_m, err)
        else:
            err = ((got - want).abs() / (0.06 * want + 1e-4)).max().item()
            worst_v = max(worst_v, err)
    print('worst mean error %.3f, worst variance error %.3f of the tolerance' % (worst_m, worst_v))
    assert worst_m <= 1.0 and worst_v <= 1.0


class _Stack(torch.nn.Module):
    """A short graph in the models.yolo.Model form the recalibrator walks: a module list whose members carry `f`."""

    def __init__(self):
        super().__init__()
        from yolo_amd.models import common
        self.model = torch.nn.Sequential(common.Conv(3, 16, 3, 2), common.BottleneckCSP(16, 40, n=2), common.SPP(40, 32, (5, 9, 13)),
                                         common.Bottleneck(32, 32, True, e=0.5), common.Conv(32, 21, 3, 2))
        for m in self.model:
            m.f = -1

    def forward(self, x):
        return self.model(x)


def test_shallow_stack_statistics_are_tight():
    from yolo_amd.adaptive_bn import Recalibrator
    from yolo_amd.utils import torch_utils
    torch.manual_seed(3)
    model = _Stack()
    torch_utils.initialize_weights(model)
    util.randomize_bn(model, 9)
    bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    for m in bns:
        m.momentum = 1.0
    x = torch.randint(0, 256, (8, 3, 96, 80), generator=torch.Generator().manual_seed(2), dtype=torch.uint8)
    ref_model = __import__('copy').deepcopy(model).train()
    with torch.no_grad():
        ref_model(x.float() / 255.0)
    ref = ref_model.state_dict()
    model = model.cuda().train()
    rec = Recalibrator(model)
    rec.step(x.cuda())
    rec.finish()
    sd = model.state_dict()
    worst_m = worst_v = 0.0
    for k in _bn_keys(sd):
        got, want = sd[k].cpu().double(), ref[k].double()
        if k.endswith('running_mean'):
            std = ref[k.replace('running_mean', 'running_var')].double().sqrt()
            worst_m = max(worst_m, ((got - want).abs() / (0.002 * std + 5e-4)).max().item())
        else:
            worst_v = max(worst_v, ((got - want).abs() / (0.004 * want + 1e-4)).max().item())
    print('worst mean error %.3f, worst variance error %.3f of the tolerance' % (worst_m, worst_v))
    assert len(bns) == 13 and worst_m <= 1.0 and worst_v <= 1.0


def test_deepcopy_leaves_the_input_model_alone():
    from yolo_amd.adaptive_bn import adaptive_bn
    model = util.build_model('yolov5s').cuda()
    before = {k: v.clone() for k, v in model.state_dict().items()}
    x = torch.randint(0, 256, (2, 3, 64, 64), dtype=torch.uint8)
    out = adaptive_bn(model, [x], deepcp=True)
    assert out is not model
    assert all(torch.equal(v, before[k]) for k, v in model.state_dict().items())
    assert any(not torch.equal(v.cpu(), before[k].cpu()) for k, v in out.state_dict().items() if k.endswith('running_mean'))


def test_fused_model_is_refused():
    from yolo_amd.adaptive_bn import Recalibrator
    from yolo_amd.engine import Unsupported
    model = util.quiet(util.build_model('yolov5s').fuse).cuda()
    with pytest.raises(Unsupported):
        Recalibrator(model).step(torch.zeros(1, 3, 64, 64, dtype=torch.uint8, device='cuda'))


@pytest.mark.parametrize('shape', [(2, 32, 32, 40, 24, 3, 1), (3, 64, 104, 20, 20, 1, 1), (1, 128, 256, 20, 12, 3, 2), (2, 24, 13, 17, 9, 3, 1)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_conv_epilogue_statistics(shape):
    """y5_conv_stats_f16: the same output as y5_conv_f16 (bit for bit, same kernel) and, summed over the partial rows,
    the per-channel sums of that fp16 output - for every tile shape that accumulates, into a column window of a wider
    statistics buffer that already holds another tensor's sums."""
    from tests.test_gpu_conv import _desc
    L, lib = _lib()
    B, Ci, Co, H, W, k, s = shape
    g = torch.Generator().manual_seed(sum(shape))
    st = L.current_stream()
    cip, cop = (Ci + 7) // 8 * 8, (Co + 7) // 8 * 8
    x = torch.zeros(B, H, W, cip, dtype=torch.float16, device='cuda')
    x[..., :Ci] = torch.randn(B, H, W, Ci, generator=g).half().cuda()
    w = (torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5).half().cuda()
    rows, off, ctot = 8, 16, cop + 24
    taken = 0
    for v in range(lib.y5_conv_num_variants() + 1):
        d, y, keep = _desc('cuda', x, w, stride=s)
        d.variant = v
        if lib.y5_conv_f16(C.byref(d), st) != 0:
            continue
        torch.cuda.synchronize()
        ref = y.clone()
        y.fill_(float('nan'))
        part = torch.zeros(rows, 2 * ctot, dtype=torch.float32, device='cuda')
        part[:, :off] = 5.0   # somebody else's columns
        rc = lib.y5_conv_stats_f16(C.byref(d), part.data_ptr(), rows, ctot, off, st)
        if rc == L.EINVAL:
            continue          # a kernel family that does not accumulate
        L.check(rc)
        torch.cuda.synchronize()
        taken += 1
        assert torch.equal(y, ref)
        yd = ref.double().reshape(-1, cop)
        sums = part.double().sum(0)
        assert torch.allclose(sums[off:off + cop], yd.sum(0), rtol=1e-5, atol=1e-3 * yd.abs().max().item())
        assert torch.allclose(sums[ctot + off:ctot + off + cop], (yd * yd).sum(0), rtol=1e-5, atol=1e-3)
        assert bool((part[:, :off] == 5.0).all()) and bool((part[:, off + cop:ctot] == 0).all())
        assert bool((part[:, ctot:ctot + off] == 0).all()) and bool((part[:, ctot + off + cop:] == 0).all())
    assert taken >= 10
