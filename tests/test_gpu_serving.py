"""The serving loop that allocates nothing: launch plans writing into a ring of output tensors (engine.set_output_ring)
and NMS into caller-held tensors (nms_padded(out=...), PaddedNms), against the default path that returns fresh tensors."""
import pytest
import torch

import util
from yolo_amd import _lib, engine
from yolo_amd.nms import PaddedNms, nms_buffers, nms_padded

pytestmark = pytest.mark.gpu


def _model(dev):
    model = util.build_model('yolov5s')
    util.randomize_bn(model, 3)
    model.eval()
    util.quiet(model.fuse)
    return model.half().to(dev)


def _allocations(dev):
    return torch.cuda.memory_stats(dev)['allocation.all.allocated']


def test_ring_outputs_and_prepared_nms_equal_the_default_path(dev):
    model = _model(dev)
    g = torch.Generator().manual_seed(8)
    batches = [torch.rand(4, 3, 128, 160, generator=g).half().to(dev) for _ in range(3)]
    want = []
    for x in batches:                      # default path: fresh tensors every call
        z, xs = model(x)
        want.append((z.clone(), [t.clone() for t in xs]) + tuple(t.clone() for t in nms_padded(z, 0.001, 0.6, return_index=True)))
    engine.set_output_ring(model, 2)
    B, N, no = want[0][0].shape
    post = PaddedNms(B, N, no, torch.float16, dev, conf_thres=0.001, iou_thres=0.6, ring=2)
    for x in batches:                      # warm: ring tensors exist, every lazy table of torch is built
        post(model(x)[0])
    torch.cuda.synchronize()
    calls = engine.calls(model)
    for step in range(10):
        x = batches[step % 3]
        z, xs = model(x)
        det, count, kept, ncand = post(z)
        if step >= 7:                      # results are read before their ring slot comes round again
            zr, xr, dr, cr, kr, nr = want[step % 3]
            assert torch.equal(z, zr) and all(torch.equal(a, b) for a, b in zip(xs, xr))
            assert torch.equal(count, cr) and torch.equal(ncand, nr)
            for b in range(B):
                n = int(count[b])
                assert torch.equal(det[b, :n], dr[b, :n]) and torch.equal(kept[b, :n], kr[b, :n])
    torch.cuda.synchronize()
    assert engine.calls(model) == calls + 10, 'every forward must have run on the HIP engine'


def test_the_step_itself_leaves_the_allocation_counter_alone(dev):
    model = _model(dev)
    x = torch.rand(2, 3, 96, 96, generator=torch.Generator().manual_seed(1)).half().to(dev)
    engine.set_output_ring(model, 2)
    z, _ = model(x)
    post = PaddedNms(*z.shape, torch.float16, dev, conf_thres=0.01, ring=2)
    for _ in range(3):
        post(model(x)[0])
    torch.cuda.synchronize()
    before = _allocations(dev)
    for _ in range(20):
        post(model(x)[0])
    torch.cuda.synchronize()
    assert _allocations(dev) == before
    outs = {model(x)[0].data_ptr() for _ in range(6)}
    assert len(outs) == 2, 'a ring of two sets of outputs'
    engine.set_output_ring(model, 0)       # back to fresh tensors: two live results never share memory
    a, b = model(x)[0], model(x)[0]
    assert a.data_ptr() != b.data_ptr()
    assert _allocations(dev) > before


def test_nms_padded_into_caller_tensors_equals_the_allocating_call(dev):
    g = torch.Generator().manual_seed(4)
    pred = torch.rand(3, 2000, 85, generator=g)
    pred[..., :4] *= 300
    pred = pred.half().to(dev)
    ref = nms_padded(pred, 0.3, 0.5, return_index=True)
    out = nms_buffers(3, device=dev)
    before = _allocations(dev)
    got = nms_padded(pred, 0.3, 0.5, return_index=True, check_overflow=False, out=out)
    assert _allocations(dev) == before
    assert all(a is b for a, b in zip(got, out))
    assert torch.equal(got[1], ref[1]) and torch.equal(got[3], ref[3])
    for b in range(3):
        n = int(ref[1][b])
        assert n > 0 and torch.equal(got[0][b, :n], ref[0][b, :n]) and torch.equal(got[2][b, :n], ref[2][b, :n])
    # class filter and agnostic mode through the prepared form
    post = PaddedNms(3, 2000, 85, torch.float16, dev, conf_thres=0.3, iou_thres=0.5, classes=[0, 17, 42], agnostic=True)
    ref = nms_padded(pred, 0.3, 0.5, classes=[0, 17, 42], agnostic=True, return_index=True)
    got = post(pred)
    assert torch.equal(got[1], ref[1])
    for b in range(3):
        n = int(ref[1][b])
        assert torch.equal(got[0][b, :n], ref[0][b, :n]) and torch.equal(got[2][b, :n], ref[2][b, :n])


def test_requests_the_caller_tensors_cannot_serve_are_refused(dev):
    pred = torch.rand(2, 500, 85, generator=torch.Generator().manual_seed(2)).half().to(dev)
    out = nms_buffers(2, device=dev)
    with pytest.raises(_lib.Y5Error, match='more than max_candidates'):
        nms_padded(pred, 0.001, 0.6, max_candidates=64, out=out)          # overflow: no silent retry with out=
    with pytest.raises(ValueError):
        nms_padded(pred, 0.3, 0.6, out=nms_buffers(3, device=dev))        # wrong batch
    with pytest.raises(ValueError):
        nms_padded(pred, 0.3, 0.6, out=out[:2])
    big = torch.zeros(64, 8, 85, dtype=torch.float16, device=dev)
    with pytest.raises(_lib.Y5Error, match='built-in scratch'):
        nms_padded(big, 0.3, 0.6, max_candidates=65536, out=nms_buffers(64, device=dev))
    post = PaddedNms(64, 8, 85, torch.float16, dev, max_candidates=65536)  # the prepared form brings its own workspace then
    assert not post.builtin and int(post(big)[1].sum()) == 0
    with pytest.raises(ValueError):
        post(big[:32])


def test_bf16_predictions_are_read_in_place_and_equal_their_fp32_copy(dev):
    g = torch.Generator().manual_seed(6)
    pred = torch.rand(2, 3000, 85, generator=g)
    pred[..., :4] *= 400
    pred = pred.bfloat16().to(dev)
    ref = nms_padded(pred.float(), 0.6, 0.6, return_index=True)
    got = nms_padded(pred, 0.6, 0.6, return_index=True)
    assert all(torch.equal(a, b) for a, b in zip(got, ref)) and int(ref[1].min()) > 0
    post = PaddedNms(2, 3000, 85, torch.bfloat16, dev, conf_thres=0.6, iou_thres=0.6)
    assert int(ref[3].max()) < 65536  # no image overflows the default capacity, which PaddedNms does not grow
    assert all(torch.equal(a, b) for a, b in zip(post(pred), ref))


def test_graphed_forward_replays_the_plan_and_refuses_a_changed_model(dev):
    model = _model(dev)
    g = torch.Generator().manual_seed(9)
    a, b = (torch.rand(2, 3, 96, 128, generator=g).half().to(dev) for _ in range(2))
    za, zb = model(a)[0].clone(), model(b)[0].clone()
    x = a.clone()
    fwd = engine.GraphedForward(model, x, sets=2)
    calls = engine.calls(model)
    z1, xs1 = fwd()
    x.copy_(b)
    z2, _ = fwd()
    assert z1.data_ptr() != z2.data_ptr() and torch.equal(z1, za) and torch.equal(z2, zb)
    x.copy_(a)
    z3, xs3 = fwd()
    assert z3.data_ptr() == z1.data_ptr() and torch.equal(z3, za) and len(xs3) == 3
    assert engine.calls(model) == calls + 3
    with torch.no_grad():
        model.model[0].conv.conv.weight.mul_(1.25)
    with pytest.raises(_lib.Y5Error, match='capture it again'):
        fwd()


def test_pipeline_with_three_steps_in_flight_equals_the_serial_path_and_allocates_nothing(dev):
    from yolo_amd.serving import Pipeline
    model = _model(dev)
    g = torch.Generator().manual_seed(12)
    batches = [torch.rand(3, 3, 128, 96, generator=g).half().to(dev) for _ in range(5)]
    want = []
    for x in batches:
        z, xs = model(x)
        want.append((z.clone(),) + tuple(t.clone() for t in nms_padded(z, 0.01, 0.6, return_index=True)))
    pipe = Pipeline(model, batches[0], depth=3, ring=2, conf_thres=0.01, iou_thres=0.6)
    eng = engine.engine_of(model)
    assert len(next(iter(eng.lanes.values()))) == 2, 'two more launch plans beside the model\'s own'
    for i in range(6):                     # warm every lane and ring slot
        pipe(batches[i % 5])
    pipe.synchronize()
    before, calls = _allocations(dev), engine.calls(model)
    steps = [pipe(batches[i % 5]) for i in range(6)]       # six steps: every (lane, slot) pair once, none overwritten yet
    assert len({s.stream for s in steps}) == 3 and len({s.z.data_ptr() for s in steps}) == 6
    assert _allocations(dev) == before and engine.calls(model) == calls + 6
    for i, s in enumerate(steps):
        s.synchronize()
        zr, dr, cr, kr, nr = want[i % 5]
        assert torch.equal(s.z, zr) and torch.equal(s.count, cr) and torch.equal(s.n_candidates, nr)
        for b in range(3):
            n = int(cr[b])
            assert torch.equal(s.det[b, :n], dr[b, :n]) and torch.equal(s.kept_index[b, :n], kr[b, :n])
    # a change of the model drops every lane's plan; the pipeline's next step builds them again and follows the change
    with torch.no_grad():
        model.model[0].conv.conv.weight.mul_(0.5)
    z_new = model(batches[1])[0].clone()
    assert not eng.lanes
    s = pipe(batches[1]).synchronize()
    s = pipe(batches[1]).synchronize()     # a lane other than 0
    assert torch.equal(s.z, z_new) and not torch.equal(s.z, want[1][0])
    with pytest.raises(ValueError):
        pipe(batches[0][:2])


def test_pipeline_copy_input_and_forward_only(dev):
    from yolo_amd.serving import Pipeline
    model = _model(dev)
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(3)).half().to(dev)
    ref = model(x)[0].clone()
    pipe = Pipeline(model, x, depth=2, ring=1, copy_input=True, nms=False)
    buf = x.clone()
    s = pipe(buf)
    buf.zero_()                            # the pipeline read its own copy
    s.wait()
    assert s.det is None and torch.equal(s.z, ref)
