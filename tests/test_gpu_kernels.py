"""Data-movement kernels, Detect decode and NMS through the C ABI: bit-exact against torch / golden / oracle."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import util
from oracle import nms_c, y5ref
from yolo_amd import _lib, nms as hip_nms
from yolo_amd.utils import utils as yu

pytestmark = pytest.mark.gpu


def nhwc(t, ld, dev):
    B, Cc, H, W = t.shape
    out = torch.zeros(B, H, W, ld, dtype=torch.float16, device=dev)
    out[..., :Cc] = t.permute(0, 2, 3, 1)
    return out


@pytest.mark.parametrize('focus', [0, 1])
def test_pack_image(dev, focus):
    lib = _lib.lib()
    x = torch.rand(3, 3, 32, 48, generator=torch.Generator().manual_seed(0)).half().to(dev)
    ld = 16 if focus else 8
    y = torch.full((3, 32 // (1 + focus), 48 // (1 + focus), ld), float('nan'), dtype=torch.float16, device=dev)
    _lib.check(lib.y5_pack_image_f16(x.data_ptr(), y.data_ptr(), 3, 3, 32, 48, focus, ld, _lib.current_stream()))
    ref = y5ref.focus_gather(x) if focus else x
    assert torch.equal(y[..., :ref.shape[1]], ref.permute(0, 2, 3, 1)) and bool((y[..., ref.shape[1]:] == 0).all())


@pytest.mark.parametrize('k', [5, 9, 13])
def test_maxpool_and_cascade(dev, k):
    lib = _lib.lib()
    x = torch.randn(2, 16, 20, 14, generator=torch.Generator().manual_seed(k)).half().to(dev)
    xn = nhwc(x, 16, dev)
    y = torch.zeros_like(xn)
    _lib.check(lib.y5_maxpool_f16(xn.data_ptr(), 16, y.data_ptr(), 16, 2, 20, 14, 16, k, _lib.current_stream()))
    assert torch.equal(y.permute(0, 3, 1, 2), F.max_pool2d(x, k, 1, k // 2))
    if k == 5:  # SPP in one launch, written into slices 1..3 of a 4-slice concat buffer
        cat = torch.zeros(2, 20, 14, 64, dtype=torch.float16, device=dev)
        cat[..., :16] = xn
        _lib.check(lib.y5_maxpool_cascade_f16(cat.data_ptr(), 64, cat.data_ptr() + 32, 64, 16, 3, 2, 20, 14, 16, 5,
                                               _lib.current_stream()))
        ref = torch.cat([x] + [F.max_pool2d(x, kk, 1, kk // 2) for kk in (5, 9, 13)], 1)
        assert torch.equal(cat.permute(0, 3, 1, 2), ref)


@pytest.mark.parametrize('c,h,w,levels,k', [(32, 20, 20, 3, 5), (24, 20, 14, 3, 5), (16, 33, 40, 2, 3), (8, 45, 45, 3, 5),
                                            (256, 20, 20, 3, 5), (16, 7, 5, 4, 3),
                                            # the in-register cascade (k = 5, <= 3 levels, planes up to 20 x 20, c % 32 == 0)
                                            (64, 16, 20, 3, 5), (32, 20, 13, 2, 5), (32, 5, 3, 1, 5), (96, 1, 20, 3, 5)])
def test_maxpool_cascade_shapes(dev, c, h, w, levels, k):
    """Two vectors per workgroup (c % 16 == 0), one vector, and planes too large for the three-plane kernels."""
    lib = _lib.lib()
    x = torch.randn(2, c, h, w, generator=torch.Generator().manual_seed(c + h)).half().to(dev)
    ld = (levels + 1) * c
    cat = torch.zeros(2, h, w, ld, dtype=torch.float16, device=dev)
    cat[..., :c] = x.permute(0, 2, 3, 1)
    _lib.check(lib.y5_maxpool_cascade_f16(cat.data_ptr(), ld, cat.data_ptr() + 2 * c, ld, c, levels, 2, h, w, c, k, _lib.current_stream()))
    ref = torch.cat([x] + [F.max_pool2d(x, kk, 1, kk // 2) for kk in [l * (k - 1) + k for l in range(levels)]], 1)
    assert torch.equal(cat.permute(0, 3, 1, 2), ref)


@pytest.mark.parametrize('mode', ['nearest', 'bilinear'])
def test_upsample_into_concat_slice(dev, mode):
    lib = _lib.lib()
    z = util.golden('block_upsample_%s_concat' % mode)
    a, b = torch.from_numpy(z['a']).half().to(dev), torch.from_numpy(z['b']).half().to(dev)
    an = nhwc(a, 16, dev)
    cat = torch.zeros(2, 12, 10, 40, dtype=torch.float16, device=dev)
    cat[..., 16:] = b.permute(0, 2, 3, 1)
    _lib.check(lib.y5_upsample2x_f16(an.data_ptr(), 16, cat.data_ptr(), 40, 2, 6, 5, 16, 0 if mode == 'nearest' else 1,
                                      _lib.current_stream()))
    ref = torch.cat([F.interpolate(a.float(), scale_factor=2.0, mode=mode, **({} if mode == 'nearest' else {'align_corners': False})).half(), b], 1)
    got = cat.permute(0, 3, 1, 2)
    if mode == 'nearest':
        assert torch.equal(got, ref)
    else:  # fp32 blend rounded once: at most one fp16 ulp from torch's own kernel
        assert float((got.float() - ref.float()).abs().max()) <= 2e-3 * float(ref.float().abs().max())


def test_copy_channels(dev):
    lib = _lib.lib()
    src = torch.randn(50, 24, generator=torch.Generator().manual_seed(0)).half().to(dev)
    dst = torch.zeros(50, 40, dtype=torch.float16, device=dev)
    _lib.check(lib.y5_copy_channels_f16(src.data_ptr(), 24, dst.data_ptr() + 16, 40, 50, 24, _lib.current_stream()))
    assert torch.equal(dst[:, 8:32], src) and bool((dst[:, :8] == 0).all()) and bool((dst[:, 32:] == 0).all())


def decode_level(dev, raw, stride, anchors_wh, z_out, row0, no):
    lib = _lib.lib()
    B, c, ny, nx = raw.shape
    na = c // no
    ld = (c + 7) // 8 * 8
    head = nhwc(raw, ld, dev)
    xo = torch.zeros(B, na, ny, nx, no, dtype=torch.float16, device=dev)
    d = _lib.DecodeDesc()
    d.head, d.head_ld, d.x_out, d.z_out = head.data_ptr(), ld, xo.data_ptr(), z_out.data_ptr()
    d.batch, d.ny, d.nx, d.na, d.no = B, ny, nx, na, no
    d.z_rows_total, d.z_row0, d.stride = z_out.shape[1], row0, float(stride)
    for a in range(na):
        d.anchor_wh[2 * a], d.anchor_wh[2 * a + 1] = float(anchors_wh[a, 0]), float(anchors_wh[a, 1])
    _lib.check(lib.y5_detect_decode_f16(C.byref(d), _lib.current_stream()))
    torch.cuda.synchronize()
    return xo


@pytest.mark.parametrize('tag', ['nc80_64x96', 'nc5_128x128', 'nc1_odd'])
def test_decode_bit_exact_against_reference_fixture(dev, tag):
    z = util.golden('decode_' + tag)
    zg = torch.from_numpy(z['z'])
    zo = torch.zeros(zg.shape, dtype=torch.float16, device=dev)
    row0 = 0
    for i in range(3):
        raw = torch.from_numpy(z['raw%d' % i]).to(dev)
        xo = decode_level(dev, raw, z['stride'][i], z['anchor_grid'][i].reshape(-1, 2), zo, row0, zg.shape[2])
        assert torch.equal(xo.cpu(), torch.from_numpy(z['x%d' % i]))
        row0 += 3 * raw.shape[2] * raw.shape[3]
    assert torch.equal(zo.cpu().view(torch.int16), zg.view(torch.int16))


@pytest.mark.parametrize('tag', ['nc80_64x96', 'nc5_128x128', 'nc1_odd'])
def test_decode_all_levels_in_one_launch_against_reference_fixture(dev, tag):
    """y5_detect_decode_multi_f16 (what the engine records) must give the fixture's z and x tensors bit for bit."""
    lib = _lib.lib()
    z = util.golden('decode_' + tag)
    zg = torch.from_numpy(z['z'])
    no = zg.shape[2]
    zo = torch.zeros(zg.shape, dtype=torch.float16, device=dev)
    levels = (_lib.DecodeDesc * 3)()
    keep, xs, row0 = [], [], 0
    for i in range(3):
        raw = torch.from_numpy(z['raw%d' % i]).to(dev)
        B, c, ny, nx = raw.shape
        na, ld = c // no, (c + 7) // 8 * 8
        head = nhwc(raw, ld, dev)
        xo = torch.zeros(B, na, ny, nx, no, dtype=torch.float16, device=dev)
        keep.append(head)
        xs.append(xo)
        d = levels[i]
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
        assert torch.equal(xs[i].cpu(), torch.from_numpy(z['x%d' % i]))
    assert torch.equal(zo.cpu().view(torch.int16), zg.view(torch.int16))
    assert lib.y5_detect_decode_multi_f16(levels, 5, _lib.current_stream()) == _lib.EINVAL


def test_decode_full_size_level_against_oracle(dev):
    g = torch.Generator().manual_seed(0)
    raw = (torch.randn(2, 255, 80, 80, generator=g) * 4).half()
    ag = torch.tensor([10., 13, 16, 30, 33, 23]).view(1, 1, 3, 1, 1, 2)
    zr, xr = y5ref.detect_decode([raw], ag, torch.tensor([8.]), True)
    zo = torch.zeros(zr.shape, dtype=torch.float16, device=dev)
    xo = decode_level(dev, raw.to(dev), 8.0, ag.view(3, 2), zo, 0, 85)
    assert torch.equal(xo.cpu(), xr[0]) and torch.equal(zo.cpu().view(torch.int16), zr.view(torch.int16))


def test_normalize_u8_equals_half_then_divide(dev):
    """y5_normalize_u8_f16 against torch's CPU `img.half() / 255.0` (test.py:152-154): every byte value, and a tensor
    whose size is not a multiple of the 16-byte vector."""
    from yolo_amd import preprocess
    allv = torch.arange(256, dtype=torch.int32).to(torch.uint8)
    g = torch.Generator().manual_seed(2)
    odd = torch.randint(0, 256, (3, 3, 37, 41), generator=g, dtype=torch.int32).to(torch.uint8)
    for t in (allv, odd):
        got = preprocess.normalize(t.to(dev))
        ref = t.half() / 255.0
        assert got.dtype is torch.float16 and got.shape == t.shape
        assert torch.equal(got.cpu().view(torch.int16), ref.view(torch.int16))


def test_sigmoid_of_every_fp16_value(dev):
    """The decode kernel's 4-instruction sigmoid against the libm/IEEE-division form and against torch's CPU half
    sigmoid (the op the reference applies, models/yolo.py:41) on all 65536 bit patterns."""
    lib = _lib.lib()
    bits = torch.arange(65536, dtype=torch.int32).to(torch.int16)
    x = bits.view(torch.float16).to(dev)
    fast, slow = torch.empty_like(x), torch.empty_like(x)
    _lib.check(lib.y5_sigmoid_f16(x.data_ptr(), fast.data_ptr(), 65536, 0, _lib.current_stream()))
    _lib.check(lib.y5_sigmoid_f16(x.data_ptr(), slow.data_ptr(), 65536, 1, _lib.current_stream()))
    torch.cuda.synchronize()
    ref = torch.sigmoid(bits.view(torch.float16).float()).half()
    finite = ~torch.isnan(ref)
    assert torch.equal(fast.cpu().view(torch.int16)[finite], slow.cpu().view(torch.int16)[finite])
    assert torch.equal(fast.cpu().view(torch.int16)[finite], ref.view(torch.int16)[finite])
    assert bool(torch.isnan(fast.cpu()[~finite]).all())


NMS_CASES = ['multilabel', 'conf01', 'ties', 'agnostic', 'classes', 'over300', 'nc1', 'empty', 'half_input', 'single']


@pytest.mark.parametrize('name', NMS_CASES)
def test_nms_bit_exact_against_reference_fixture(dev, name):
    z = util.golden('nms_' + name)
    classes = list(z['classes']) if len(z['classes']) else None
    pred = torch.from_numpy(z['pred']).to(dev)
    det, cnt, kept, ncand = hip_nms.nms_padded(pred, float(z['conf']), float(z['iou']), classes=classes,
                                               agnostic=bool(z['agnostic']), return_index=True)
    assert np.array_equal(cnt.cpu().numpy(), z['count']) and np.array_equal(det.cpu().numpy(), z['det'])
    ref = nms_c.non_max_suppression(z['pred'], float(z['conf']), float(z['iou']), classes=classes, agnostic=bool(z['agnostic']))
    for b, (rdet, ridx, rn) in enumerate(ref):
        assert int(ncand[b]) == rn and np.array_equal(kept[b, :len(ridx)].cpu().numpy(), ridx)
        assert bool((kept[b, len(ridx):] == -1).all())
    out = yu.non_max_suppression(pred, float(z['conf']), float(z['iou']), classes=classes, agnostic=bool(z['agnostic']))
    for b, o in enumerate(out):
        assert (o is None) == (z['count'][b] == 0)
        if o is not None:
            assert o.is_cuda and np.array_equal(o.cpu().numpy(), z['det'][b, :z['count'][b]])


def synth_prediction(seed, B, N, nc, frac=0.05):
    g = torch.Generator().manual_seed(seed)
    cxy = torch.rand(B, N, 2, generator=g) * 640
    wh = torch.exp(torch.randn(B, N, 2, generator=g) * 0.5 + 3.5)
    obj = torch.rand(B, N, 1, generator=g) * (torch.rand(B, N, 1, generator=g) < frac)
    cls = torch.rand(B, N, nc, generator=g) ** 3
    return torch.cat((cxy, wh, obj, cls), 2)


def test_nms_full_size_indices_equal_the_oracle(dev):
    """25200 rows x 80 classes per image in fp16, conf 0.001 / iou 0.65 (the test.py setting): kept candidate
    indices and rows must equal the C oracle's exactly; also exercises the capacity-retry path."""
    pred = synth_prediction(3, 3, 25200, 80).half()
    ref = nms_c.non_max_suppression(pred.numpy(), 0.001, 0.65)
    det, cnt, kept, ncand = hip_nms.nms_padded(pred.to(dev), 0.001, 0.65, return_index=True, max_candidates=4096)
    for b, (rdet, ridx, rn) in enumerate(ref):
        assert rn > 4096 and int(ncand[b]) == rn and int(cnt[b]) == len(ridx) == 300
        assert np.array_equal(kept[b, :300].cpu().numpy(), ridx) and np.array_equal(det[b, :300].cpu().numpy(), rdet)
    scores = det[..., 4].cpu()
    assert bool((scores[:, :-1] >= scores[:, 1:]).all())  # descending confidence


def test_nms_large_candidate_sets_sort_through_global_memory(dev):
    pred = synth_prediction(4, 1, 25200, 80, frac=0.3).half()   # > 16384 candidates: multi-block sort
    ref = nms_c.non_max_suppression(pred.numpy(), 0.05, 0.5)
    det, cnt, kept, ncand = hip_nms.nms_padded(pred.to(dev), 0.05, 0.5, return_index=True, max_candidates=1 << 20)
    assert ref[0][2] > 16384 and int(ncand[0]) == ref[0][2]
    assert np.array_equal(kept[0, :len(ref[0][1])].cpu().numpy(), ref[0][1])


def test_nms_sort_paths_by_candidate_count(dev):
    """One batch whose images have 0 .. 20000 candidates: the sort picks its path per image on the device (nothing,
    one chunk, several chunks merged through global memory, the one-workgroup fallback above 16384 keys)."""
    counts = [0, 1, 2, 63, 2047, 2048, 2049, 4097, 9000, 16384, 16385, 20000]
    N = 20000
    g = torch.Generator().manual_seed(11)
    pred = synth_prediction(11, len(counts), N, 1, frac=1.0)
    pred[..., 4] = torch.rand(len(counts), N, generator=g) * 0.5 + 0.5
    pred[..., 5] = torch.randint(20, 200, (len(counts), N), generator=g).float() / 200   # many exact score ties
    for b, c in enumerate(counts):
        pred[b, torch.randperm(N, generator=g)[c:], 4] = 0.0
    ref = nms_c.non_max_suppression(pred.numpy(), 0.01, 0.45)
    det, cnt, kept, ncand = hip_nms.nms_padded(pred.to(dev), 0.01, 0.45, return_index=True, max_candidates=32768)
    for b, (rdet, ridx, rn) in enumerate(ref):
        assert rn == counts[b] == int(ncand[b])
        assert int(cnt[b]) == len(ridx) and np.array_equal(kept[b, :len(ridx)].cpu().numpy(), ridx)
        assert np.array_equal(det[b, :len(ridx)].cpu().numpy(), rdet)


def test_scale_detections_equals_the_reference_statements(dev):
    """y5_scale_coords_f32 against scale_coords + clip_coords (utils/utils.py:193-214) executed by torch on the CPU,
    per image, on boxes that reach beyond the letterbox padding; with and without ratio_pad and detect.py's round."""
    g = torch.Generator().manual_seed(8)
    img1 = (384, 640)
    img0 = [(480, 640, 3), (1080, 1920, 3), (333, 517, 3), (640, 384, 3)]
    B, max_det = len(img0), 300
    det = torch.zeros(B, max_det, 6)
    det[..., :2] = torch.rand(B, max_det, 2, generator=g) * 700 - 30
    det[..., 2:4] = det[..., :2] + torch.rand(B, max_det, 2, generator=g) * 300
    det[..., 4] = torch.rand(B, max_det, generator=g)
    count = torch.tensor([300, 17, 0, 255], dtype=torch.int32)
    rps = [None, ((0.333, 0.333), (0.5, 12.0)), None, None]
    for rnd in (False, True):
        ref = det.clone()
        for b in range(B):
            c = ref[b, :count[b], :4]          # a view: the statements below are the reference's, in place
            if rps[b] is None:
                gain = min(img1[0] / img0[b][0], img1[1] / img0[b][1])
                pad = (img1[1] - img0[b][1] * gain) / 2, (img1[0] - img0[b][0] * gain) / 2
            else:
                gain, pad = rps[b][0][0], rps[b][1]
            c[:, [0, 2]] -= pad[0]
            c[:, [1, 3]] -= pad[1]
            c[:, :4] /= gain
            c[:, 0].clamp_(0, img0[b][1]); c[:, 1].clamp_(0, img0[b][0])
            c[:, 2].clamp_(0, img0[b][1]); c[:, 3].clamp_(0, img0[b][0])
            if rnd:
                ref[b, :count[b], :4] = c.round()
        got = hip_nms.scale_detections(det.clone().to(dev), count.to(dev), img1, img0, rps, round_=rnd)
        assert torch.equal(got.cpu().view(torch.int32), ref.view(torch.int32))


@pytest.mark.parametrize('name', ['a', 'b', 'c'])
def test_match_predictions_equals_the_reference_fixture(dev, name):
    """y5_match_predictions_f32 (clip + best-IoU matching of a whole batch in one launch) against the fixture computed
    with the reference's box_iou / xywh2xyxy / clip_coords: correct flags and the clipped boxes bit for bit."""
    from yolo_amd import metrics
    z = util.golden('metrics_match_' + name)
    det = torch.from_numpy(z['det']).to(dev)
    count = torch.from_numpy(z['count']).to(dev)
    correct = metrics.match_predictions(det, count, torch.from_numpy(z['targets']), float(z['height']), float(z['width']),
                                        torch.from_numpy(z['iouv']), check_overflow=True)
    assert np.array_equal(correct.cpu().numpy(), z['correct'])
    for b in range(det.shape[0]):
        n = int(z['count'][b])
        assert np.array_equal(det[b, :n].cpu().numpy().view(np.int32), z['clipped'][b, :n].view(np.int32))
    stats = metrics.batch_stats(det, count, correct, torch.from_numpy(z['targets']))
    assert sum(s[0].shape[0] for s in stats) == int(z['count'].sum())


def test_match_predictions_large_random_batch_against_oracle(dev):
    from oracle import metrics_ref
    from yolo_amd import metrics
    g = torch.Generator().manual_seed(21)
    B, h, w = 16, 640, 640
    labels = []
    det = torch.zeros(B, 300, 6)
    count = torch.randint(0, 301, (B,), generator=g, dtype=torch.int32)
    for b in range(B):
        nl = int(torch.randint(0, 90, (1,), generator=g))
        cxy, wh = torch.rand(nl, 2, generator=g) * 0.8 + 0.1, torch.rand(nl, 2, generator=g) * 0.3 + 0.02
        cls = torch.randint(0, 6, (nl, 1), generator=g).float()
        labels.append(torch.cat((torch.full((nl, 1), float(b)), cls, cxy, wh), 1))
        n = int(count[b])
        if nl and n:
            t = torch.randint(0, nl, (n,), generator=g)
            c = torch.cat((cxy[t] - wh[t] / 2, cxy[t] + wh[t] / 2), 1) * torch.tensor([w, h, w, h])
            det[b, :n, :4] = c + (torch.rand(n, 4, generator=g) - 0.5) * 40
            det[b, :n, 5] = torch.where(torch.rand(n, generator=g) < 0.8, cls[t, 0], torch.randint(0, 6, (n,), generator=g).float())
        det[b, :n, 4] = torch.sort(torch.rand(n, generator=g), descending=True).values
    targets = torch.cat(labels, 0)
    iouv = torch.linspace(0.5, 0.95, 10)
    ref_det = det.clone()
    ref = torch.zeros(B, 300, 10, dtype=torch.bool)
    for b in range(B):
        n = int(count[b])
        ref[b, :n] = metrics_ref.match_image(ref_det[b, :n], targets[targets[:, 0] == b, 1:], float(h), float(w), iouv)
    d = det.to(dev)
    got = metrics.match_predictions(d, count.to(dev), targets, h, w, iouv)
    assert torch.equal(got.cpu(), ref) and bool(ref.any())
    assert torch.equal(d.cpu().view(torch.int32), ref_det.view(torch.int32))


def test_box_nms_primitive_and_merge_path(dev):
    g = torch.Generator().manual_seed(0)
    xy = torch.rand(3000, 2, generator=g) * 300
    boxes = torch.cat((xy, xy + torch.rand(3000, 2, generator=g) * 60 + 5), 1)
    scores = (torch.randint(0, 100, (3000,), generator=g) / 100).float()
    keep = hip_nms.nms(boxes.to(dev), scores.to(dev), 0.5, max_keep=1024).cpu().numpy()
    assert np.array_equal(keep, nms_c.nms(boxes.numpy(), scores.numpy(), 0.5)[:1024])
    assert hip_nms.nms(boxes[:0].to(dev), scores[:0].to(dev), 0.5).numel() == 0
    z = util.golden('nms_merge')
    out = yu.non_max_suppression(torch.from_numpy(z['pred']).to(dev), float(z['conf']), float(z['iou']), merge=True)
    for b, o in enumerate(out):
        assert o.shape[0] == z['count'][b]
        assert np.allclose(o.cpu().numpy(), z['det'][b, :z['count'][b]], rtol=1e-4, atol=1e-3)


def test_nms_is_idempotent_on_its_own_output(dev):
    """Property: boxes that survived do not suppress each other, so feeding them back keeps all of them."""
    pred = synth_prediction(5, 1, 5000, 4, frac=0.5).to(dev)
    det, cnt = hip_nms.nms_padded(pred, 0.1, 0.5, agnostic=True)
    n = int(cnt[0])
    d = det[0, :n]
    keep = hip_nms.nms(d[:, :4].contiguous(), d[:, 4].contiguous(), 0.5, max_keep=300)
    assert keep.tolist() == list(range(n))


@pytest.mark.parametrize('shape', [(2, 128, 16, 24), (3, 256, 20, 20), (1, 72, 7, 9), (2, 72, 8, 8), (9, 128, 80, 80), (5, 256, 40, 40), (3, 128, 20, 12),
                                   # the persistent kernel on head widths inside a power-of-two layout (yolov5m: 192, 384; yolov5x: 320)
                                   (3, 192, 40, 24), (2, 320, 20, 20), (5, 384, 12, 20), (4, 192, 80, 80)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_detect_level_in_one_kernel_equals_conv_then_decode(dev, shape):
    """y5_detect_f16 (head conv with the decode as its epilogue) against y5_conv_f16 followed by y5_detect_decode_f16 on
    the same tensors: z and the permuted raw copy must agree -- identical wherever the two convolutions round alike, and
    within the conv tolerance everywhere (the tile shapes, hence the fp32 summation orders, differ)."""
    lib = _lib.lib()
    B, Ci, ny, nx = shape
    na, no = 3, 85
    cout, cpad = na * no, 256
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(B, ny, nx, Ci, generator=g).half().to(dev)
    w = torch.zeros(256, (Ci + 63) // 64 * 64, dtype=torch.float16, device=dev)
    w[:cout, :Ci] = (torch.randn(cout, Ci, generator=g) / Ci ** 0.5).half().to(dev)
    bias = torch.zeros(cpad, device=dev)
    bias[:cout] = torch.randn(cout, generator=g).to(dev)
    rows_total = na * ny * nx + 40          # the level sits at row offset 40 of a longer z
    st = _lib.current_stream()

    cd = _lib.ConvDesc()
    cd.x, cd.w, cd.bias = x.data_ptr(), w.data_ptr(), bias.data_ptr()
    cd.batch, cd.h, cd.w_in, cd.cin, cd.x_ld = B, ny, nx, Ci, Ci
    cd.ho, cd.wo, cd.cout, cd.y_ld = ny, nx, cpad, cpad
    cd.ksize, cd.stride, cd.pad, cd.kpad, cd.w_rows = 1, 1, 0, w.shape[1], 256
    cd.act, cd.slope = _lib.ACT_NONE, 0.0

    def dec_desc(x_out, z_out, head=None):
        d = _lib.DecodeDesc()
        d.x_out, d.z_out = x_out.data_ptr(), z_out.data_ptr()
        if head is not None:
            d.head, d.head_ld = head.data_ptr(), cpad
        d.batch, d.ny, d.nx, d.na, d.no = B, ny, nx, na, no
        d.z_rows_total, d.z_row0, d.stride = rows_total, 40, 16.0
        for a, (aw, ah) in enumerate([(10., 13.), (16., 30.), (33., 23.)]):
            d.anchor_wh[2 * a], d.anchor_wh[2 * a + 1] = aw, ah
        return d

    head = torch.zeros(B, ny, nx, cpad, dtype=torch.float16, device=dev)
    cd.y = head.data_ptr()
    _lib.check(lib.y5_conv_f16(C.byref(cd), st))
    x_ref = torch.zeros(B, na, ny, nx, no, dtype=torch.float16, device=dev)
    z_ref = torch.zeros(B, rows_total, no, dtype=torch.float16, device=dev)
    _lib.check(lib.y5_detect_decode_f16(C.byref(dec_desc(x_ref, z_ref, head)), st))
    cd.y = None
    taken = 0
    for v in range(lib.y5_detect_num_variants() + 1):
        x_got = torch.full_like(x_ref, float('nan'))
        z_got = torch.zeros_like(z_ref)
        rc = lib.y5_detect_f16(C.byref(cd), C.byref(dec_desc(x_got, z_got)), v, st)
        if (ny * nx * no) % 8:
            assert rc == _lib.EINVAL, 'runs that are not 16-byte aligned belong to the two-kernel path'
            continue
        persistent = v == lib.y5_detect_num_variants()  # the last variant: compiled for these input widths
        if persistent and (Ci not in (128, 192, 256, 320, 384) or no < 16):
            assert rc == _lib.EINVAL
            taken += 1
            continue
        _lib.check(rc)
        torch.cuda.synchronize()
        taken += 1
        assert not bool(torch.isnan(x_got).any())
        assert bool((z_got[:, :40] == 0).all()), 'rows before the level must stay untouched'
        for got, ref in ((x_got, x_ref), (z_got, z_ref)):
            d = (got.float() - ref.float()).abs()
            assert bool((d <= 2e-3 * ref.float().abs() + 2e-3 * ref.float().abs().max()).all()), float(d.max())
        assert float((x_got == x_ref).float().mean()) > 0.97
    assert taken == (0 if (ny * nx * no) % 8 else lib.y5_detect_num_variants() + 1)


@pytest.mark.parametrize('nc', [1, 2, 3])
def test_decode_small_class_counts_on_aligned_levels(dev, nc):
    """nc = 1 / 2 (6 / 7 outputs per anchor) on levels whose runs are 16-byte aligned, so that the vector store path
    runs (the reference fixture for nc = 1 has odd sizes and takes the scalar path): one launch per level and all levels
    in one launch against the oracle's half-mode decode, and the one-kernel Detect level against the two-kernel path."""
    lib = _lib.lib()
    no, na, B = nc + 5, 3, 2
    g = torch.Generator().manual_seed(nc)
    sizes, strides = [(16, 16), (8, 8), (4, 4)], [8., 16., 32.]
    raws = [(torch.randn(B, na * no, ny, nx, generator=g) * 3).half() for ny, nx in sizes]
    ag = torch.tensor([[10., 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]).view(3, 1, 3, 1, 1, 2)
    zr, xr = y5ref.detect_decode(raws, ag, torch.tensor(strides), True)
    assert (zr.shape[1] * no) % 8 == 0
    zo = torch.zeros(zr.shape, dtype=torch.float16, device=dev)
    row0 = 0
    for i, raw in enumerate(raws):
        xo = decode_level(dev, raw.to(dev), strides[i], ag[i].view(3, 2), zo, row0, no)
        assert torch.equal(xo.cpu(), xr[i])
        row0 += na * raw.shape[2] * raw.shape[3]
    assert torch.equal(zo.cpu().view(torch.int16), zr.view(torch.int16))
    # all levels in one launch
    zo2 = torch.zeros_like(zo)
    levels = (_lib.DecodeDesc * 3)()
    keep, row0 = [], 0
    for i, raw in enumerate(raws):
        ld = (na * no + 7) // 8 * 8
        head = nhwc(raw.to(dev), ld, dev)
        xo = torch.zeros(B, na, raw.shape[2], raw.shape[3], no, dtype=torch.float16, device=dev)
        keep += [head, xo]
        d = levels[i]
        d.head, d.head_ld, d.x_out, d.z_out = head.data_ptr(), ld, xo.data_ptr(), zo2.data_ptr()
        d.batch, d.ny, d.nx, d.na, d.no = B, raw.shape[2], raw.shape[3], na, no
        d.z_rows_total, d.z_row0, d.stride = zr.shape[1], row0, strides[i]
        for a in range(na):
            d.anchor_wh[2 * a], d.anchor_wh[2 * a + 1] = float(ag[i, 0, a, 0, 0, 0]), float(ag[i, 0, a, 0, 0, 1])
        row0 += na * raw.shape[2] * raw.shape[3]
    _lib.check(lib.y5_detect_decode_multi_f16(levels, 3, _lib.current_stream()))
    torch.cuda.synchronize()
    assert torch.equal(zo2, zo)


@pytest.mark.parametrize('nc', [1, 2])
def test_small_class_count_model_runs_on_the_engine(dev, nc):
    """A whole nc = 1 / 2 graph at a size whose levels are aligned (Detect levels as one kernel each): against the oracle."""
    import os
    import yaml
    from yolo_amd import engine
    model = util.build_model('yolov5s', nc=nc)
    util.randomize_bn(model, 11)
    model = model.eval().half().to(dev)
    x = torch.rand(2, 3, 128, 128, generator=torch.Generator().manual_seed(5)).half()
    z, xs = model(x.to(dev))
    assert engine.calls(model) == 1 and z.shape == (2, 3 * (256 + 64 + 16), nc + 5)
    cfgd = yaml.safe_load(open(os.path.join(util.CFG_DIR, 'yolov5s.yaml')))
    cfgd['nc'] = nc
    sd = {k: v.float().cpu() for k, v in model.state_dict().items()}
    zr, xr = y5ref.forward(cfgd, sd, x.float(), precision='half', strides=model.stride)
    for got, ref in list(zip(xs, xr)) + [(z, zr)]:
        d = (got.float().cpu() - ref.float()).abs()
        assert bool((d <= 4e-3 * ref.float().abs() + 4e-3 * ref.float().abs().max()).all()), float(d.max())
    assert float((z.cpu() == zr).float().mean()) > 0.9


@pytest.mark.parametrize('agnostic', [False, True])
def test_merge_step_kernel_against_the_oracle(dev, agnostic):
    """merge=True on the GPU is one kernel after the selection; the numpy oracle (pinned to the reference's output by
    tests/golden/nms_merge.npz) is the check.  Images: a dense one (merging and redundancy both act), one with a
    single candidate and an empty one (no merge: n <= 1), one with more than 3000 candidates (no merge: n >= 3E3).
    Counts, confidences and classes are exact; merged corners agree to the summation order of the reference's fp32
    matrix product."""
    dense = synth_prediction(21, 1, 3000, 4, frac=0.25)
    dense[..., :2] = dense[..., :2] * 0.3 + 200          # crowd the centres: most kept boxes have overlapping candidates
    single = torch.zeros(1, 3000, 9)
    single[0, 7] = torch.tensor([100., 120., 40., 30., 0.9, 0.05, 0.8, 0.05, 0.05])
    empty = torch.zeros(1, 3000, 9)
    many = synth_prediction(22, 1, 3000, 4, frac=0.9)
    pred = torch.cat((dense, single, empty, many), 0)
    ref = y5ref.non_max_suppression(pred.numpy(), 0.1, 0.5, agnostic=agnostic, merge=True, nms_fn=nms_c.nms)
    plain = y5ref.non_max_suppression(pred.numpy(), 0.1, 0.5, agnostic=agnostic, nms_fn=nms_c.nms)
    det, cnt, kept, ncand = hip_nms.nms_padded(pred.to(dev), 0.1, 0.5, agnostic=agnostic, merge=True, return_index=True)
    assert 1 < int(ncand[0]) < 3000 and int(ncand[1]) == 1 and int(ncand[2]) == 0 and int(ncand[3]) >= 3000
    assert len(ref[0][0]) < len(plain[0][0]), 'the redundancy filter must drop something in the dense image'
    assert not np.array_equal(ref[0][0][:, :4], plain[0][0][:len(ref[0][0]), :4])
    assert np.array_equal(ref[3][0], plain[3][0]) and len(ref[2][0]) == 0
    for b, (rdet, rflat, rn) in enumerate(ref):
        n = len(rdet)
        assert int(cnt[b]) == n and int(ncand[b]) == rn
        got = det[b, :n].cpu().numpy()
        assert np.array_equal(got[:, 4:], rdet[:, 4:])
        assert np.allclose(got[:, :4], rdet[:, :4], rtol=1e-5, atol=1e-3)
        assert np.array_equal(kept[b, :n].cpu().numpy(), rflat)          # kept_index follows the compaction
        assert bool((det[b, n:] == 0).all()) and bool((kept[b, n:] == -1).all())
    # the mirror's CPU stage list (the reference's own statements) agrees with the oracle on the same input
    host = yu._nms_stage_list(pred.clone(), 0.1, 0.5, True, None, agnostic, yu._greedy_nms_host)
    for b, h in enumerate(host):
        assert (h is None and len(ref[b][0]) == 0) or np.allclose(h.numpy(), ref[b][0], rtol=1e-5, atol=1e-3)
    out = yu.non_max_suppression(pred.to(dev), 0.1, 0.5, merge=True, agnostic=agnostic)   # the list API takes the same path
    assert out[2] is None and all(torch.equal(o.cpu(), det[b, :o.shape[0]].cpu()) for b, o in enumerate(out) if o is not None)
