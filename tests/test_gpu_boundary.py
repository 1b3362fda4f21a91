"""The callers' side of the path on the GPU: test-time augmentation, checkpoints (attempt_load / Ensemble), copies of a
model that has already run on the engine, and the end-to-end agreement of the fp16 HIP path with the fp32 oracle."""
import copy
import io
import os

import numpy as np
import pytest
import torch
import yaml

import util
from oracle import nms_c, y5ref
from yolo_amd import engine

pytestmark = pytest.mark.gpu


def _cfg(name):
    return yaml.safe_load(open(os.path.join(util.CFG_DIR, name + '.yaml')))


def test_test_time_augmentation_against_the_oracle(dev):
    """Model.forward(x, augment=True) (models/yolo.py:89-103) with the three views on the engine, against the oracle's
    half-precision restatement.  The scaled views are resized by torch's bilinear kernel on the GPU and by its CPU kernel
    in the oracle, which may differ in the last fp16 bit of some input pixels; the bound is the whole-graph one."""
    model = util.build_model('yolov5s')
    util.randomize_bn(model, 11)
    model = model.eval().half().to(dev)
    x = torch.from_numpy(util.golden('tta_yolov5s')['x']).half()
    y, none = model(x.to(dev), augment=True)
    assert none is None and engine.calls(model) == 3
    sd = {k: v.float().cpu() for k, v in model.state_dict().items()}
    want = y5ref.forward_augment(_cfg('yolov5s'), sd, x.float(), precision='half', strides=model.stride)
    assert y.shape == want.shape and y.dtype is torch.float16
    d = (y.float().cpu() - want.float()).abs()
    tol = 4e-3 * want.float().abs() + 4e-3 * want.float().abs().max()
    assert bool((d <= tol).all()), 'max |d| %g' % float(d.max())
    assert float((d == 0).float().mean()) >= 0.80
    # and against the reference's own fp32 output of the same input, as an fp16-vs-fp32 sanity bound on the scores
    ref32 = torch.from_numpy(util.golden('tta_yolov5s')['y'])
    assert float((y.float().cpu()[..., 4:] - ref32[..., 4:]).abs().max()) < 0.05


def _save_like_the_reference(model, path):
    """torch.save({'model': model}) with the classes pickled under the reference's module paths (models.yolo.Model,
    models.common.Conv ...), which is what a checkpoint written by the reference's train scripts contains."""
    from yolo_amd import dropin
    from yolo_amd.models import common, yolo
    dropin.install()
    classes = [c for mod in (common, yolo) for c in vars(mod).values()
               if isinstance(c, type) and c.__module__ in (common.__name__, yolo.__name__)]
    old = {c: c.__module__ for c in classes}
    try:
        for c in classes:
            c.__module__ = c.__module__.replace('yolo_amd.', '')
        torch.save({'epoch': -1, 'model': model, 'optimizer': None}, path)
    finally:
        for c, m in old.items():
            c.__module__ = m
    with open(path, 'rb') as f:
        blob = f.read()
    assert b'yolo_amd' not in blob


def test_checkpoint_round_trip_through_attempt_load(dev, tmp_path):
    """models/experimental.py:128-141: a whole-module checkpoint under the reference's import paths loads through
    attempt_load (fp32, fused, eval), runs on the engine and gives the tensors of the directly built model; two
    checkpoints give an Ensemble whose output is the mean of the two."""
    from yolo_amd.models.experimental import Ensemble, attempt_load
    direct = []
    paths = []
    for seed in (0, 1):
        m = util.build_model('yolov5s', seed=seed)
        util.randomize_bn(m, 11 + seed)
        m.names = ['c%d' % i for i in range(80)]
        p = str(tmp_path / ('m%d.pt' % seed))
        _save_like_the_reference(m, p)
        paths.append(p)
        direct.append(util.quiet(copy.deepcopy(m).float().fuse).eval().half().to(dev))
    x = torch.rand(2, 3, 64, 96, generator=torch.Generator().manual_seed(8)).half().to(dev)
    one = util.quiet(attempt_load, paths[0], map_location='cpu')
    assert type(one).__name__ == 'Model' and not one.training and all(c.bn is None for c in one.modules() if type(c).__name__ == 'Conv')
    one = one.half().to(dev)
    z, xs = one(x)
    z0, xs0 = direct[0](x)
    assert engine.calls(one) == 1 and torch.equal(z, z0) and all(torch.equal(a, b) for a, b in zip(xs, xs0))
    ens = util.quiet(attempt_load, paths, map_location='cpu')
    assert isinstance(ens, Ensemble) and len(ens) == 2 and ens.names == one.names and torch.equal(ens.stride, one.stride.cpu())
    ens = ens.half().to(dev)
    y, none = ens(x)
    z1, _ = direct[1](x)
    assert none is None and all(engine.calls(m) == 1 for m in ens)
    assert torch.equal(y, torch.stack([z0, z1]).mean(0))


def test_a_model_that_ran_on_the_engine_can_be_copied_and_saved(dev):
    """The reference's workflows deep-copy and pickle models after evaluating them (ModelEMA, the pruning search,
    torch.save of ckpt['model'], adaptive_BN's deepcopy): the engine keeps its plans outside the module."""
    model = util.build_model('yolov5s')
    util.randomize_bn(model, 2)
    model = model.eval().half().to(dev)
    x = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(1)).half().to(dev)
    z0, _ = model(x)
    assert engine.calls(model) == 1
    clone = copy.deepcopy(model)
    buf = io.BytesIO()
    torch.save(model, buf)
    buf.seek(0)
    loaded = torch.load(buf, weights_only=False)
    for m in (clone, loaded):
        assert engine.calls(m) == 0
        z, _ = m(x)
        assert engine.calls(m) == 1 and torch.equal(z, z0)
    assert '_y5_engine' not in model.__dict__ and all('engine' not in k for k in model.__dict__)


def test_plan_cache_is_bounded(dev):
    """A rectangular-batch loader feeds many shapes: the engine keeps the most recently used plans only."""
    model = util.build_model('yolov5s').eval().half().to(dev)
    shapes = [(1, 3, 64, 64), (1, 3, 64, 96), (1, 3, 96, 64), (2, 3, 64, 64), (1, 3, 96, 96), (1, 3, 128, 64)]
    first = None
    for s in shapes:
        z, _ = model(torch.rand(*s, generator=torch.Generator().manual_seed(3)).half().to(dev))
        first = z if first is None else first
    eng = engine.engine_of(model)
    assert len(eng.plans) == engine.MAX_PLANS < len(shapes)
    z, _ = model(torch.rand(*shapes[0], generator=torch.Generator().manual_seed(3)).half().to(dev))  # evicted: rebuilt
    assert torch.equal(z, first) and engine.calls(model) == len(shapes) + 1


# ------------------------------------------------------------------------------------------------------------------
def _as_targets(dets):
    """Oracle detections [(det (n,6) xyxy conf cls, ...)] -> the loader's `targets` rows (image, class, xywh in pixels)."""
    rows = []
    for b, det in enumerate(dets):
        for x1, y1, x2, y2, conf, cls in det:
            rows.append([b, cls, (x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1])
    return torch.tensor(rows, dtype=torch.float32).reshape(-1, 6)


def _map50(det, count, targets, dev, size=640):
    """(mAP@0.5, share of the labels matched by a detection of their class with IoU > 0.5) of padded detections against
    `targets` (pixel xywh), with the evaluation loop's own matching (test.py:220-247) and AP computation
    (utils/utils.py:217-306)."""
    from yolo_amd import metrics
    iouv = torch.tensor([0.5])
    t = targets.clone()
    t[:, 2:] /= size                      # the matcher takes normalised labels and scales them by (width, height)
    correct = metrics.match_predictions(det.clone(), count, t, size, size, iouv)
    stats = metrics.batch_stats(det, count, correct, t)
    tp, conf, pcls, tcls = [np.concatenate([np.asarray(s[i]) for s in stats], 0) for i in range(4)]
    p, r, ap, f1, _ = metrics.ap_per_class(tp, conf, pcls, tcls)
    return float(ap[:, 0].mean()), float(tp.sum()) / max(1, t.shape[0])


def _padded(dets, dev):
    det = torch.zeros(len(dets), 300, 6)
    cnt = torch.zeros(len(dets), dtype=torch.int32)
    for b, dd in enumerate(dets):
        det[b, :len(dd)] = torch.from_numpy(dd)
        cnt[b] = len(dd)
    return det.to(dev), cnt.to(dev)


def test_end_to_end_agreement_with_the_fp32_oracle(dev):
    """The fp16 HIP path end to end (conv -> decode -> NMS) against the reference's fp32 CPU arithmetic (the oracle in
    fp32 + its C NMS) on the benchmark's calibrated synthetic yolov5s (head gain 2), 8 images of 640x640, conf 0.005 /
    iou 0.5 (100-300 detections per image); the fp32 oracle's detections serve as labels.

    A half-precision graph of 60 layers with random weights is chaotic end to end: the reference's OWN half path (the
    oracle's precision='half' restatement) reaches mAP@0.5 ~0.7 against its fp32 path, and max |z_fp16 - z_fp32| is
    hundreds of pixels on a few boxes, so no elementwise bound against fp32 means anything.  What can be stated, and is
    asserted here: the HIP path is as close to the fp32 result as the reference's half path is -
        |mAP@0.5(HIP vs fp32) - mAP@0.5(half oracle vs fp32)| <= 0.05, the same for the matched-label share,
    and no further from it than torch-ROCm's own fp16 kernels on the same modules; mAP@0.5(HIP) >= 0.6.
    Measured: HIP 0.688 / 0.746 matched, half oracle 0.705 / 0.764, torch-ROCm fp16 0.380 / 0.489, fp32 against itself
    0.960 / 0.967."""
    from yolo_amd.nms import nms_padded
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    conf, iou, B = 0.005, 0.5, 8
    model = util.build_model('yolov5s')
    util.calibrate(model, seed=1, size=640, head_gain=2.0)
    fused = util.quiet(model.fuse).eval()
    sd32 = {k: v.clone() for k, v in fused.state_dict().items()}
    gpu = copy.deepcopy(fused).half().to(dev)
    sd16 = {k: v.float().cpu() for k, v in gpu.state_dict().items()}
    x = torch.rand(B, 3, 640, 640, generator=torch.Generator().manual_seed(1)).half()
    cfg = _cfg('yolov5s')
    z32, _ = y5ref.forward(cfg, sd32, x.float(), precision='fp32', strides=fused.stride)
    ref = [r[0] for r in nms_c.non_max_suppression(z32.numpy(), conf, iou)]
    assert sum(len(r) for r in ref) >= 100 * B, 'the synthetic head must produce a real detection load'
    targets = _as_targets(ref)

    zh, _ = gpu(x.to(dev))
    assert engine.calls(gpu) == 1
    hip = _map50(*nms_padded(zh, conf, iou), targets, dev)
    with engine.disabled():
        zt, _ = gpu(x.to(dev))
    eager = _map50(*nms_padded(zt, conf, iou), targets, dev)
    z16, _ = y5ref.forward(cfg, sd16, x.float(), precision='half', strides=fused.stride)
    half = _map50(*_padded([r[0] for r in nms_c.non_max_suppression(z16.numpy(), conf, iou)], dev), targets, dev)
    same = _map50(*_padded(ref, dev), targets, dev)
    print('vs fp32 oracle detections (mAP@0.5, matched share): HIP fp16 %.4f %.4f | half oracle %.4f %.4f | torch-ROCm fp16 '
          '%.4f %.4f | fp32 oracle itself %.4f %.4f' % (hip + half + eager + same))
    assert same[0] > 0.95 and same[1] > 0.95   # not 1.0: overlapping same-class labels compete for one detection each
    assert abs(hip[0] - half[0]) <= 0.05 and abs(hip[1] - half[1]) <= 0.05
    assert hip[0] >= eager[0] - 0.05 and hip[1] >= eager[1] - 0.05
    assert hip[0] >= 0.6
