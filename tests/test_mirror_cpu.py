"""The host-side mirror of the reference interface (yolo_amd.models / yolo_amd.utils) on the CPU, against the
fixtures generated from the imported reference."""
import os
import sys

import numpy as np
import pytest
import torch

import util
from yolo_amd import dropin, engine
from yolo_amd.utils import utils as yu


@pytest.mark.parametrize('name', util.BLOCKS)
def test_block_module_matches_reference(name):
    m, x, y = util.block_from_fixture(name)
    with torch.no_grad():
        assert torch.equal(m(x), y)


@pytest.mark.parametrize('cfg', ['yolov5s', 'yolov5m', 'yolov5l', 'yolov5x', 'yolov5s_ori'])
def test_seeded_model_equals_reference_model(cfg):
    """Same yaml + same seed must give the reference's parameter shapes, save list, strides, anchors and the very
    same random initial weights (checksum), i.e. modules are created in the reference's order."""
    z = util.golden('graph_' + cfg)
    model = util.build_model(cfg)
    sd = {k: v for k, v in model.state_dict().items() if 'num_batches_tracked' not in k}
    assert list(sd) == z['keys'].tolist()
    assert [';'.join(map(str, v.shape)) for v in sd.values()] == z['shapes'].tolist()
    assert model.save == z['save'].tolist()
    assert np.array_equal(model.stride.numpy(), z['stride'])
    assert np.array_equal(model.model[-1].anchors.numpy(), z['anchors'])
    assert np.array_equal(model.model[-1].anchor_grid.numpy(), z['anchor_grid'])
    checksum = sum(float(v.double().abs().sum()) for v in model.state_dict().values() if v.dtype.is_floating_point)
    assert abs(checksum - float(z['checksum'])) <= 1e-9 * float(z['checksum'])


@pytest.mark.parametrize('cfg', ['yolov5s', 'yolov5s_ori'])
def test_forward_and_fuse_match_reference(cfg):
    z = util.golden('e2e_' + cfg)
    model = util.build_model(cfg)
    util.randomize_bn(model, 11)
    model.eval()
    with torch.no_grad():
        for tag in ('a', 'b'):
            zz, xs = model(torch.from_numpy(z['x_' + tag]))
            assert torch.equal(zz, torch.from_numpy(z['z_' + tag]))
            for i, t in enumerate(xs):
                assert torch.equal(t, torch.from_numpy(z['x%d_%s' % (i, tag)]))
        fused = util.quiet(model.fuse)
        assert fused is model and all(m.bn is None for m in model.modules() if type(m).__name__ == 'Conv')
        zf, _ = model(torch.from_numpy(z['x_a']))
        assert torch.equal(zf, torch.from_numpy(z['z_a_fused']))


def test_train_mode_returns_raw_maps_and_cpu_is_never_eligible():
    model = util.build_model('yolov5s')
    out = model.train()(torch.zeros(1, 3, 64, 64))
    assert isinstance(out, list) and [tuple(t.shape) for t in out] == [(1, 3, 8, 8, 85), (1, 3, 4, 4, 85), (1, 3, 2, 2, 85)]
    model.eval()
    with torch.no_grad():
        assert not engine.eligible(model, torch.zeros(1, 3, 64, 64))
        assert not engine.eligible(model, torch.zeros(1, 3, 64, 64).half())


NMS_CASES = ['multilabel', 'conf01', 'ties', 'agnostic', 'classes', 'over300', 'nc1', 'empty', 'half_input', 'single']


@pytest.mark.parametrize('name', NMS_CASES)
def test_nms_on_host_tensors_matches_reference(name):
    z = util.golden('nms_' + name)
    classes = list(z['classes']) if len(z['classes']) else None
    out = yu.non_max_suppression(torch.from_numpy(z['pred']), float(z['conf']), float(z['iou']), classes=classes,
                                 agnostic=bool(z['agnostic']))
    assert len(out) == len(z['count'])
    for b, o in enumerate(out):
        if z['count'][b] == 0:
            assert o is None
        else:
            assert o.dtype is torch.float32 and np.array_equal(o.numpy(), z['det'][b, :z['count'][b]])


def test_merge_nms_matches_reference():
    z = util.golden('nms_merge')
    out = yu.non_max_suppression(torch.from_numpy(z['pred']), float(z['conf']), float(z['iou']), merge=True)
    for b, o in enumerate(out):
        assert o.shape[0] == z['count'][b]
        assert np.allclose(o.numpy(), z['det'][b, :z['count'][b]], rtol=1e-5, atol=1e-4)


def test_box_helpers():
    b = torch.tensor([[10., 20., 4., 6.], [0., 0., 2., 2.]])
    xyxy = yu.xywh2xyxy(b)
    assert xyxy.tolist() == [[8., 17., 12., 23.], [-1., -1., 1., 1.]]
    assert torch.equal(yu.xyxy2xywh(xyxy), b)
    assert np.array_equal(yu.xywh2xyxy(b.numpy()), xyxy.numpy())
    iou = yu.box_iou(xyxy, xyxy)
    assert torch.allclose(iou.diag(), torch.ones(2)) and float(iou[0, 1]) == 0.0
    c = torch.tensor([[-5., 10., 700., 500.]])
    yu.clip_coords(c, (480, 640))
    assert c.tolist() == [[0., 10., 640., 480.]]
    s = yu.scale_coords((320, 640), torch.tensor([[64., 32., 576., 288.]]), (240, 480))
    assert torch.allclose(s, torch.tensor([[48., 24., 432., 216.]]))
    assert yu.make_divisible(33, 8) == 40 and yu.check_img_size(641, 32) == 672


def test_dropin_aliases_reference_module_paths():
    saved = {k: sys.modules.get(k) for k in ('models', 'models.yolo', 'models.common', 'utils', 'utils.utils')}
    try:
        for k in saved:
            sys.modules.pop(k, None)
        dropin.install()
        from models.yolo import Model, Detect  # noqa: F401
        from models.common import Conv, BottleneckCSP  # noqa: F401
        from utils.utils import non_max_suppression
        import yolo_amd.models.yolo as y
        assert Model is y.Model and non_max_suppression is yu.non_max_suppression
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_pruned_shapes_run_on_the_module_walk():
    model = util.prune_like(util.build_model('yolov5s_ori'))
    z = util.golden('graph_yolov5s_ori_pruned_a0')
    with torch.no_grad():
        out = model.eval()(torch.zeros(1, 3, 64, 64))[0]
    assert list(out.shape) == z['out_shape'].tolist()
    odd = [m.out_channels for m in model.modules() if isinstance(m, torch.nn.Conv2d) and m.out_channels % 8]
    assert len(odd) > 40  # the plan really has irregular widths


def test_adaptive_bn_has_no_cpu_path():
    """The recalibration forward runs on the HIP kernels only: a model on the CPU is refused, not silently run elsewhere."""
    import pytest
    from yolo_amd.adaptive_bn import Recalibrator, adaptive_bn
    model = util.build_model('yolov5s')
    with pytest.raises(RuntimeError):
        Recalibrator(model)
    with pytest.raises(RuntimeError):
        adaptive_bn(model, [torch.zeros(1, 3, 64, 64, dtype=torch.uint8)])


def test_checkpoint_with_the_reference_module_paths_loads_through_attempt_load(tmp_path):
    """models/experimental.py:128-141 on the CPU: a whole-module checkpoint whose pickle names models.yolo.Model /
    models.common.* (what the reference's training scripts write) resolves to the mirror classes, comes back fp32, fused
    and in eval mode, and computes what the directly fused model computes; two files give the mean Ensemble."""
    import copy
    from test_gpu_boundary import _save_like_the_reference
    from yolo_amd.models.experimental import Ensemble, attempt_load
    paths, direct = [], []
    for seed in (0, 1):
        m = util.build_model('yolov5s', seed=seed)
        util.randomize_bn(m, 11 + seed)
        m.names = ['c%d' % i for i in range(80)]
        p = str(tmp_path / ('m%d.pt' % seed))
        _save_like_the_reference(m.half(), p)   # strip_optimizer stores half checkpoints (utils/utils.py:648-654)
        paths.append(p)
        direct.append(util.quiet(copy.deepcopy(m).float().fuse).eval())
    x = torch.rand(1, 3, 64, 96, generator=torch.Generator().manual_seed(8))
    with torch.no_grad():
        one = util.quiet(attempt_load, paths[0], map_location='cpu')
        assert type(one).__name__ == 'Model' and not one.training and next(one.parameters()).dtype is torch.float32
        assert torch.equal(one(x)[0], direct[0](x)[0])
        ens = util.quiet(attempt_load, paths, map_location='cpu')
        assert isinstance(ens, Ensemble) and len(ens) == 2 and ens.names == one.names
        y, none = ens(x)
        assert none is None and torch.equal(y, torch.stack([direct[0](x)[0], direct[1](x)[0]]).mean(0))
