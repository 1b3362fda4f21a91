"""The engine's per-forward host work: a cached model state that is re-walked only when something changed.

What has to invalidate it (reference call sites): Model.fuse (models/yolo.py:155-163), .half() / .float() around the
evaluation (filter_pruning/fp_test.py:128,327), load_state_dict, structured pruning assigning new Parameters
(filter_pruning/torch_pruning/prune/structured.py:50-60), BatchNorm statistics moving in place
(filter_pruning/fp_train.py:207-261), forward hooks of the FLOP counter (filter_pruning/fp_utils/utils.py:99-103)."""
import copy

import torch
import torch.nn as nn

import util
from yolo_amd import engine


def _fresh():
    model = util.build_model('yolov5s', seed=0).eval()
    return model, engine.Engine(model)


def test_steady_state_does_not_walk_the_model():
    model, eng = _fresh()
    sig = eng.state().signature
    assert eng.walks == 1
    for _ in range(5):
        assert eng.state().signature == sig
    assert eng.walks == 1


def test_dtype_conversion_fuse_and_load_state_dict_change_the_signature():
    model, eng = _fresh()
    seen = {eng.state().signature}
    model.half()
    seen.add(eng.state().signature)
    assert eng.state().dtype is torch.float16
    model.float()
    seen.add(eng.state().signature)
    util.quiet(model.fuse)
    seen.add(eng.state().signature)
    sd = copy.deepcopy(model.state_dict())
    model.load_state_dict(sd)
    seen.add(eng.state().signature)
    assert len(seen) == 5
    walks = eng.walks
    eng.state()
    assert eng.walks == walks


def test_pruning_style_parameter_assignment_and_in_place_updates_are_seen():
    model, eng = _fresh()
    s0 = eng.state().signature
    conv = model.model[1].conv
    conv.weight = nn.Parameter(conv.weight.detach()[:40].clone())  # what prune_conv does to a layer
    conv.out_channels = 40
    s1 = eng.state().signature
    assert s1 != s0
    bn = model.model[1].bn
    bn.running_mean.add_(1.0)  # adaptive-BN moves the statistics in place
    s2 = eng.state().signature
    assert s2 != s1
    with torch.no_grad():
        model.model[0].conv.conv.weight.mul_(0.5)  # an optimiser step
    assert eng.state().signature != s2


def test_stride_attribute_replacement_is_seen():
    model, eng = _fresh()
    s0 = eng.state().signature
    model.model[-1].stride = model.model[-1].stride * 2
    st = eng.state()
    assert st.signature != s0 and st.smax == 64


def test_hooks_are_seen_without_a_walk():
    model, eng = _fresh()
    st = eng.state()
    assert not any(st.hook_tables)
    h = model.model[3].conv.register_forward_hook(lambda m, i, o: None)
    assert any(eng.state().hook_tables)
    h.remove()
    assert not any(eng.state().hook_tables)
    h = model.model[4].cv3.register_forward_pre_hook(lambda m, i: None)
    assert any(eng.state().hook_tables)
    h.remove()
    assert eng.walks == 1
    # a module added later is picked up (its assignment re-walks the model), hooks on it included
    model.model[2].extra = nn.Identity()
    model.model[2].extra.register_forward_hook(lambda m, i, o: None)
    assert any(eng.state().hook_tables) and eng.walks == 2


def test_cpu_and_ineligible_inputs_stay_on_the_torch_path():
    model, _ = _fresh()
    with torch.no_grad():
        assert engine.try_run(model, torch.zeros(1, 3, 64, 64)) is None
        assert not engine.eligible(model, torch.zeros(1, 3, 64, 64).half())


def test_an_interleaved_uint8_view_without_from_hwc_marking_is_rejected():
    import pytest
    from yolo_amd import preprocess
    model, _ = _fresh()
    frames = torch.zeros(2, 64, 64, 3, dtype=torch.uint8)
    x = preprocess.from_hwc(frames, bgr=True)
    assert engine.input_layout(x) == engine._lib.IN_U8_HWC_BGR
    assert engine.input_layout(preprocess.from_hwc(frames, bgr=False)) == engine._lib.IN_U8_HWC
    with pytest.raises(TypeError, match='from_hwc'):
        engine.input_layout(x[:1])
    with torch.no_grad():
        with pytest.raises(TypeError, match='from_hwc'):
            model(x[:1])
        with pytest.raises(TypeError, match='augment'):
            model(x, augment=True)
        z, _ = model(x)                     # marked: flipped, converted and divided like the loader's tensor
        z0, _ = model(frames.permute(0, 3, 1, 2).flip(1).contiguous())
        assert torch.equal(z, z0)


def test_only_pruned_channel_plans_are_planned_twice():
    """The second plan build (widths padded to regular ones) is for channel plans filter pruning leaves behind; the shipped
    graphs, whose widths are multiples of 16, are planned once."""
    from yolo_amd import engine
    for cfg in ('yolov5s', 'yolov5m', 'yolov5x', 'yolov5s_ori'):
        assert not engine._irregular_widths(util.build_model(cfg)), cfg
    assert engine._irregular_widths(util.prune_like(util.build_model('yolov5s_ori')))
    assert [engine.pad8(c) for c in (3, 13, 25, 55, 64, 119, 249)] == [8, 16, 32, 56, 64, 120, 256]
    engine._PAD_REGULAR[0] = True
    try:
        assert [engine.pad8(c) for c in (3, 13, 25, 55, 64, 75, 119, 249, 391)] == [8, 16, 32, 64, 64, 96, 128, 256, 512]
    finally:
        engine._PAD_REGULAR[0] = False
