"""Whole-graph HIP forward (Model.forward -> engine) + NMS against the oracle and against torch's own GPU kernels."""
import os

import numpy as np
import pytest
import torch
import yaml

import util
from oracle import nms_c, y5ref
from yolo_amd import engine
from yolo_amd.utils.utils import non_max_suppression

pytestmark = pytest.mark.gpu


def calls(model):
    return engine.calls(model)


# The bar is the tolerance.  The bit-identical fraction is a sanity floor only, set well below what is measured
# (>= 0.99 of z against the oracle): which conv tile variant the autotuner picks, and with it the fp32 summation
# order, can differ from run to run.
def agree(got, ref, frac_exact, rel=4e-3):
    """End-to-end fp16 outputs: elementwise |d| <= rel*|ref| + rel*max|ref| and a minimum share of identical values
    (half-precision graphs are chaotic end to end, so the bound is looser than per layer)."""
    got, ref = got.float().cpu(), ref.float().cpu()
    d = (got - ref).abs()
    assert bool((d <= rel * ref.abs() + rel * ref.abs().max()).all()), 'max |d| %g' % float(d.max())
    assert float((d == 0).float().mean()) >= frac_exact


@pytest.mark.parametrize('cfg,fuse,shape', [('yolov5s', False, (2, 3, 64, 64)), ('yolov5s', True, (1, 3, 128, 96)),
                                            ('yolov5s_ori', False, (2, 3, 64, 64)), ('yolov5s_ori', True, (2, 3, 96, 160)),
                                            ('yolov5m', True, (1, 3, 64, 96)), ('yolov5x', True, (1, 3, 64, 64))])
def test_forward_against_oracle(dev, cfg, fuse, shape):
    model = util.build_model(cfg)
    util.randomize_bn(model, 11)
    model.eval()
    if fuse:
        util.quiet(model.fuse)
    model = model.half().to(dev)
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(5)).half()
    z, xs = model(x.to(dev))
    assert calls(model) == 1, 'the forward must run on the HIP engine'
    cfgd = yaml.safe_load(open(os.path.join(util.CFG_DIR, cfg + '.yaml')))
    sd = {k: v.float().cpu() for k, v in model.state_dict().items()}
    zr, xr = y5ref.forward(cfgd, sd, x.float(), precision='half', strides=model.stride)
    assert z.shape == zr.shape and z.dtype is torch.float16
    for a, b in zip(xs, xr):
        assert a.shape == b.shape
        agree(a, b, 0.60)
    agree(z, zr, 0.90)
    # torch's own GPU kernels on the same modules must agree with the oracle no better than the HIP path does
    with engine.disabled():
        zt, _ = model(x.to(dev))
    assert calls(model) == 1
    agree(z, zt, 0.80)


def test_side_stream_plan_gives_the_same_tensors(dev, monkeypatch):
    """Y5_SIDE_STREAM=1 records the early Detect convolutions on the plan's second stream (y5_plan_mark_last): same
    kernels, same order per stream, so every output must be bit-identical to the single-stream plan."""
    model = util.build_model('yolov5s')
    util.randomize_bn(model, 3)
    model = model.eval().half().to(dev)
    x = torch.rand(3, 3, 96, 128, generator=torch.Generator().manual_seed(9)).half().to(dev)
    z0, xs0 = model(x)
    monkeypatch.setenv('Y5_SIDE_STREAM', '1')
    plan = engine.Plan(model, x)
    try:
        assert any('side stream' in label for label in plan.labels)
        for _ in range(3):
            z1, xs1 = plan.run(x)
        torch.cuda.synchronize()
    finally:
        plan.close()
    assert torch.equal(z0, z1) and all(torch.equal(a, b) for a, b in zip(xs0, xs1))


def test_uint8_batch_runs_on_the_engine_like_its_normalised_copy(dev):
    """A half model accepts the data loader's uint8 batch (the engine applies test.py:152-154 in its first kernel):
    the tensors of model(img.half() / 255.0), on the engine, and the torch walk offers the same convenience."""
    from yolo_amd import preprocess
    model = util.build_model('yolov5s')
    util.randomize_bn(model, 4)
    model = model.eval().half().to(dev)
    img8 = torch.randint(0, 256, (2, 3, 96, 64), generator=torch.Generator().manual_seed(1), dtype=torch.int32).to(torch.uint8).to(dev)
    z0, xs0 = model(preprocess.normalize(img8))
    n0 = calls(model)
    z1, xs1 = model(img8)
    assert calls(model) == n0 + 1, 'the uint8 forward must run on the HIP engine'
    # The uint8 first layer is bit-identical to normalise-then-convolve kernel for kernel (tests/test_gpu_conv.py); at graph
    # level the two inputs may be lowered to different kernels for layers 0 and 1 (the fp16 image can take the one-kernel
    # Focus + stride-2 conv where that measures faster, whose BatchNorm step is one fused multiply-add), so the bar here is
    # the whole-graph tolerance with nearly all values identical.
    agree(z1, z0, 0.95)
    for a, b in zip(xs1, xs0):
        agree(a, b, 0.95)
    with engine.disabled():
        zt, _ = model(img8)
    agree(z1, zt, 0.80)


@pytest.mark.parametrize('cfg', ['yolov5s', 'yolov5s_ori'])
def test_interleaved_bgr_frames_run_like_the_loaders_tensor(dev, cfg):
    """preprocess.from_hwc: the (batch, h, w, 3) BGR uint8 frames go to the model as they are; the result equals feeding
    the tensor the reference's loader builds from them (utils/datasets.py:140-141) - on the engine's fused first kernel
    for the Focus graphs (one engine call, no copy), through a converted copy for graphs that open with a plain Conv."""
    import numpy as np
    from yolo_amd import preprocess
    model = util.build_model(cfg)
    util.randomize_bn(model, 4)
    model = model.eval().half().to(dev)
    frames = torch.randint(0, 256, (2, 96, 64, 3), generator=torch.Generator().manual_seed(2), dtype=torch.int32).to(torch.uint8).numpy()
    loader = np.stack([np.ascontiguousarray(f[:, :, ::-1].transpose(2, 0, 1)) for f in frames])
    z0, xs0 = model(torch.from_numpy(loader).to(dev))
    n0 = calls(model)
    x = preprocess.from_hwc(torch.from_numpy(frames).to(dev), bgr=True)
    assert x.shape == (2, 3, 96, 64) and x.data_ptr() == x.permute(0, 2, 3, 1).data_ptr()
    z1, xs1 = model(x)
    assert calls(model) == n0 + 1
    if cfg == 'yolov5s':
        assert any('BGR HWC' in l for p in engine.engine_of(model).plans.values() for l in p.labels)
        assert torch.equal(z0, z1) and all(torch.equal(a, b) for a, b in zip(xs0, xs1))
    else:  # normalised by torch before the engine sees it: img.half() / 255.0 instead of the kernel's table, same values
        assert torch.equal(z0, z1)
    with engine.disabled():
        zt, _ = model(x)
    agree(z1, zt, 0.80)
    # a view that lost from_hwc's marking (slicing, .to(device), a collate function) is rejected, not read as RGB
    for lost in (x[:1], x.cpu().to(dev)):
        with pytest.raises(TypeError, match='from_hwc'):
            model(lost)
    with pytest.raises(TypeError, match='augment'):
        model(x, augment=True)
    zc, _ = model.float().cpu()(preprocess.from_hwc(torch.from_numpy(frames), bgr=True))   # off the GPU: flip + convert + divide
    agree(z1, zc, 0.0, rel=3e-2)


def test_detections_match_oracle_on_engine_output(dev):
    """forward -> non_max_suppression on the GPU; the NMS stage must reproduce the oracle exactly on the very
    predictions the engine produced (kept boxes are bit-identical given identical scores)."""
    model = util.build_model('yolov5s')
    util.calibrate(model, size=256, head_gain=2.0)
    model = util.quiet(model.fuse).eval().half().to(dev)
    x = torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(1)).half().to(dev)
    z, _ = model(x)
    out = non_max_suppression(z, 0.001, 0.65)
    ref = nms_c.non_max_suppression(z.cpu().numpy(), 0.001, 0.65)
    assert any(r[2] > 300 for r in ref), 'the synthetic head should produce an NMS-heavy candidate load'
    for o, (rdet, ridx, rn) in zip(out, ref):
        assert o is not None and np.array_equal(o.cpu().numpy(), rdet)


def test_pruned_channel_plan_runs_on_the_engine(dev):
    """Irregular channel counts of a filter-pruned yolov5s_ori (shapes from the reference's filter_pruning):
    every tensor is padded to a multiple of 8 channels inside the engine; results must match torch's kernels."""
    model = util.prune_like(util.build_model('yolov5s_ori'))
    util.randomize_bn(model, 3)
    model = model.eval().half().to(dev)
    x = torch.rand(2, 3, 96, 64, generator=torch.Generator().manual_seed(2)).half().to(dev)
    z, xs = model(x)
    assert calls(model) == 1
    with engine.disabled():
        zt, xt = model(x)
    for a, b in zip(xs, xt):
        agree(a, b, 0.50)
    agree(z, zt, 0.80)


def test_engine_steps_aside_for_hooks_grad_training_and_fp32(dev):
    model = util.build_model('yolov5s').eval().half().to(dev)
    x = torch.rand(1, 3, 64, 64).half().to(dev)
    model(x)
    assert calls(model) == 1
    with torch.enable_grad():
        model(x)
    assert calls(model) == 1
    h = model.model[1].conv.register_forward_hook(lambda m, i, o: None)
    model(x)
    assert calls(model) == 1
    h.remove()
    model(x)
    assert calls(model) == 2
    out = model.train()(x)
    assert isinstance(out, list) and calls(model) == 2
    model.eval()
    model.model[-1].export = True
    assert isinstance(model(x), list) and calls(model) == 2   # export mode returns the raw maps like training
    model.model[-1].export = False
    model.model[-1].training = False
    model.float()
    model(x.float())
    assert calls(model) == 2


def test_plan_follows_weight_updates_and_new_shapes(dev):
    model = util.build_model('yolov5s').eval().half().to(dev)
    x = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(0)).half().to(dev)
    z0, _ = model(x)
    with torch.no_grad():
        model.model[-1].m[0].bias.add_(1.0)
    z1, _ = model(x)
    assert not torch.equal(z0, z1), 'an in-place parameter update must rebuild the packed plan'
    with engine.disabled():
        zt, _ = model(x)
    agree(z1, zt, 0.80)
    z2, _ = model(torch.rand(3, 3, 96, 128).half().to(dev))
    assert z2.shape == (3, 3 * (12 * 16 + 6 * 8 + 3 * 4), 85) and calls(model) == 3


def test_test_time_augmentation_runs_on_the_engine(dev):
    model = util.build_model('yolov5s').eval().half().to(dev)
    x = torch.rand(1, 3, 128, 128).half().to(dev)
    y, none = model(x, augment=True)
    assert none is None and calls(model) == 3 and y.shape[0] == 1 and y.shape[2] == 85


def test_folded_upsample_concat_equals_the_materialised_plan(dev, monkeypatch):
    """nn.Upsample -> Concat read only by 1x1 convolutions is folded into those convolutions (y5_conv_cat2_f16); with
    Y5_CAT2_FUSE=0 the plan upsamples and concatenates for real.  Same arithmetic, fp32 summation order aside."""
    model = util.build_model('yolov5s')
    util.randomize_bn(model, 6)
    model = model.eval().half().to(dev)
    x = torch.rand(2, 3, 128, 192, generator=torch.Generator().manual_seed(3)).half().to(dev)
    folded = engine.Plan(model, x)
    monkeypatch.setenv('Y5_CAT2_FUSE', '0')
    plain = engine.Plan(model, x)
    try:
        assert sum('up2x' in label for label in folded.labels) == 2 and not any('upsample' in label for label in folded.labels)
        assert sum('upsample' in label for label in plain.labels) == 2 and not any('up2x' in label for label in plain.labels)
        z0, xs0 = folded.run(x)
        z1, xs1 = plain.run(x)
        torch.cuda.synchronize()
    finally:
        folded.close()
        plain.close()
    for a, b in zip(xs0, xs1):
        agree(a, b, 0.60)
    agree(z0, z1, 0.90)


def test_concat_with_a_3x3_reader_is_built(dev):
    """The fold applies only where every reader is a 1x1 convolution: a graph whose Concat feeds a 3x3 Conv keeps the
    upsample kernel and still matches torch's kernels."""
    from yolo_amd.models.yolo import Model
    cfg = {'nc': 3, 'depth_multiple': 1.0, 'width_multiple': 1.0, 'anchors': [[10, 13, 16, 30, 33, 23]],
           'backbone': [[-1, 1, 'Focus', [64, 3]], [-1, 1, 'Conv', [64, 3, 2]], [-1, 1, 'Conv', [64, 3, 2]]],
           'head': [[-1, 1, 'nn.Upsample', [None, 2, 'nearest']], [[-1, 1], 1, 'Concat', [1]], [-1, 1, 'Conv', [64, 3, 1]],
                    [[5], 1, 'Detect', ['nc', 'anchors']]]}
    torch.manual_seed(0)
    model = util.quiet(Model, cfg)
    util.randomize_bn(model, 2)
    model = model.eval().half().to(dev)
    x = torch.rand(2, 3, 64, 96, generator=torch.Generator().manual_seed(4)).half().to(dev)
    z, xs = model(x)
    assert calls(model) == 1
    plan = engine.Plan(model, x)
    try:
        assert any('upsample' in label for label in plan.labels) and not any('up2x' in label for label in plan.labels)
    finally:
        plan.close()
    with engine.disabled():
        zt, _ = model(x)
    agree(z, zt, 0.80)
    # the same graph with a 1x1 reader folds
    cfg['head'][2] = [-1, 1, 'Conv', [64, 1, 1]]
    torch.manual_seed(0)
    model = util.quiet(Model, cfg)
    util.randomize_bn(model, 2)
    model = model.eval().half().to(dev)
    z, xs = model(x)
    assert calls(model) == 1
    plan = engine.Plan(model, x)
    try:
        assert any('up2x' in label for label in plan.labels) and not any('upsample' in label for label in plan.labels)
    finally:
        plan.close()
    with engine.disabled():
        zt, _ = model(x)
    agree(z, zt, 0.80)


KNOBS = ['Y5_CHAIN_LINK', 'Y5_CHAIN_HEAD', 'Y5_CHAIN_FUSE', 'Y5_CSP_FUSE', 'Y5_CAT2_FUSE', 'Y5_DETECT_FUSE', 'Y5_STEM2_FUSE',
         'Y5_FOCUS_FUSE', 'Y5_DECODE_FUSE', 'Y5_AUTOTUNE']


@pytest.mark.parametrize('knob', KNOBS)
def test_every_fusion_knob_off_still_matches_the_oracle(dev, monkeypatch, knob):
    """Each environment switch that turns one fusion of the plan builder off must leave a correct plan: BottleneckCSPs with
    three Bottlenecks (yolov5s layers 4 and 6) exercise the chained kernels with and without the links between them."""
    model = util.build_model('yolov5s')
    util.randomize_bn(model, 11)
    model = model.eval().half().to(dev)
    x = torch.rand(2, 3, 128, 160, generator=torch.Generator().manual_seed(5)).half()
    cfgd = yaml.safe_load(open(os.path.join(util.CFG_DIR, 'yolov5s.yaml')))
    sd = {k: v.float().cpu() for k, v in model.state_dict().items()}
    zr, xr = y5ref.forward(cfgd, sd, x.float(), precision='half', strides=model.stride)
    monkeypatch.setenv(knob, '0')
    plan = engine.Plan(model, x.to(dev))
    try:
        z, xs = plan.run(x.to(dev))
        torch.cuda.synchronize()
    finally:
        plan.close()
    agree(z, zr, 0.90)
    for a, b in zip(xs, xr):
        agree(a, b, 0.60)


def test_plan_is_rebuilt_after_model_changes_and_not_otherwise(dev):
    """fuse(), a dtype round trip (filter_pruning/fp_test.py:128,327), load_state_dict and a pruning-style Parameter
    assignment each give a new plan whose output follows the change; repeated forwards walk the model once."""
    model = util.build_model('yolov5s')
    util.randomize_bn(model, 7)
    model = model.eval().half().to(dev)
    x = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(2)).half().to(dev)
    z0, _ = model(x)
    eng = engine.engine_of(model)
    plan0 = next(iter(eng.plans.values()))
    walks = eng.walks
    for _ in range(3):
        z1, _ = model(x)
    assert eng.walks == walks and next(iter(eng.plans.values())) is plan0 and torch.equal(z0, z1)

    def torch_walk():
        with engine.disabled():
            return model(x)[0]
    with torch.no_grad():
        model.model[0].conv.conv.weight.mul_(1.5)          # in place: version counter
    z2, _ = model(x)
    assert next(iter(eng.plans.values())) is not plan0 and not torch.equal(z2, z0)
    agree(z2, torch_walk(), 0.80)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    sd['model.1.conv.weight'] = sd['model.1.conv.weight'] * 0.5
    model.load_state_dict(sd)
    z3, _ = model(x)
    assert not torch.equal(z3, z2)
    agree(z3, torch_walk(), 0.80)
    model.float().half()                                    # new storage for every tensor
    z4, _ = model(x)
    assert torch.equal(z4, z3)
    model.float()
    util.quiet(model.fuse)                                  # folds in fp32, as the reference does before .half()
    model.half()
    z5, _ = model(x)
    agree(z5, torch_walk(), 0.80)
    conv = model.model[-1].m[0]
    with torch.no_grad():
        conv.weight = torch.nn.Parameter(conv.weight * 2.0)  # pruning assigns new Parameters
    z6, _ = model(x)
    assert not torch.equal(z6, z5)
    agree(z6, torch_walk(), 0.80)
    assert calls(model) == 9


def test_side_stream_nms_with_overflow_check_equals_the_serial_call(dev):
    """nms_padded(stream=side) with the default check_overflow reads the candidate totals on `side`: results bit-equal to
    the serial call, also when the first capacity is too small and the call repeats itself with a larger one."""
    from yolo_amd import nms as hip_nms
    g = torch.Generator().manual_seed(3)
    pred = torch.rand(3, 4096, 85, generator=g)
    pred[..., :4] *= 300.0
    pred = pred.half().to(dev)
    side = torch.cuda.Stream(dev)
    serial = hip_nms.nms_padded(pred, 0.3, 0.6, return_index=True)
    for cap in (None, 1000):
        got = hip_nms.nms_padded(pred, 0.3, 0.6, return_index=True, stream=side, max_candidates=cap)
        side.synchronize()
        assert int(got[3].max()) > 1000
        for a, b in zip(got, serial):
            assert torch.equal(a, b)
