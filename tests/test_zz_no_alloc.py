"""y5_nms_batched_noalloc (scratch in a device array of the library, no allocation) against y5_nms_batched (caller's
workspace) through the C ABI: identical outputs bit for bit, on two calls in a row, the second on a smaller input -
the array keeps the first call's keys and counters, so anything that must start at zero has to be cleared by the call."""
import ctypes as C

import pytest
import torch

from yolo_amd import _lib

pytestmark = pytest.mark.gpu


def _pred(batch, rows, nc, seed, dev):
    g = torch.Generator().manual_seed(seed)
    p = torch.empty(batch, rows, 5 + nc)
    p[..., 0:2] = torch.rand(batch, rows, 2, generator=g) * 640
    p[..., 2:4] = torch.exp(torch.randn(batch, rows, 2, generator=g) * 0.6 + 3.5)
    p[..., 4] = torch.rand(batch, rows, generator=g) ** 3
    p[..., 5:] = torch.rand(batch, rows, nc, generator=g) ** 6
    return p.half().to(dev)


def _run(fn, pred, cap, with_workspace, conf=0.01, iou=0.65, max_det=300):
    lib = _lib.lib()
    B, N, no = pred.shape
    dev = pred.device
    det = torch.full((B, max_det, 6), float('nan'), dtype=torch.float32, device=dev)
    count = torch.full((B,), -7, dtype=torch.int32, device=dev)
    kept = torch.full((B, max_det), -7, dtype=torch.int32, device=dev)
    ncand = torch.full((B,), -7, dtype=torch.int32, device=dev)
    d = _lib.NmsDesc()
    d.pred = pred.data_ptr()
    d.det, d.count, d.kept_index, d.n_candidates = det.data_ptr(), count.data_ptr(), kept.data_ptr(), ncand.data_ptr()
    ws = None
    if with_workspace:
        nbytes = lib.y5_nms_workspace_bytes(B, cap)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        d.workspace, d.workspace_bytes = ws.data_ptr(), nbytes
    d.iou_thres, d.conf_thres = iou, conf
    d.pred_dtype = 0
    d.batch, d.n_rows, d.no = B, N, no
    d.agnostic, d.max_candidates, d.max_det = 0, cap, max_det
    rc = fn(C.byref(d), _lib.current_stream())
    torch.cuda.synchronize()
    return rc, det, count, kept, ncand


def test_noalloc_entry_point_equals_the_workspace_one_twice_in_a_row(dev):
    lib = _lib.lib()
    torch.cuda.set_device(dev)
    big = _pred(8, 6000, 20, 1, dev)      # ~ thousands of candidates per image
    small = _pred(3, 900, 20, 2, dev)     # fewer images, fewer rows, fewer candidates: stale keys / counters would show
    for pred, cap in ((big, 65536), (small, 8192)):
        rc0, det0, cnt0, kept0, nc0 = _run(lib.y5_nms_batched, pred, cap, True)
        rc1, det1, cnt1, kept1, nc1 = _run(lib.y5_nms_batched_noalloc, pred, cap, False)
        assert rc0 == 0 and rc1 == 0, lib.y5_last_error()
        assert int(nc0.min()) > 0 and int(nc0.max()) <= cap and int(cnt0.max()) > 0
        assert torch.equal(nc0, nc1) and torch.equal(cnt0, cnt1)
        assert torch.equal(kept0, kept1) and torch.equal(det0, det1)
    assert int(_run(lib.y5_nms_batched, big, 65536, True)[2].sum()) > int(_run(lib.y5_nms_batched, small, 8192, True)[2].sum())


def test_noalloc_entry_point_refuses_what_does_not_fit(dev):
    lib = _lib.lib()
    torch.cuda.set_device(dev)
    pred = _pred(2, 64, 3, 3, dev)
    B = 64
    cap = 65536                            # 64 images x 65536 keys x 8 bytes = 32 MiB > the 16 MiB array
    assert lib.y5_nms_workspace_bytes(B, cap) > _lib.NMS_NOALLOC_BYTES
    d = _lib.NmsDesc()
    out = torch.zeros(B * 300 * 6 + B, dtype=torch.float32, device=dev)
    d.pred, d.det, d.count = pred.data_ptr(), out.data_ptr(), out.data_ptr()
    d.iou_thres, d.conf_thres, d.pred_dtype = 0.5, 0.1, 0
    d.batch, d.n_rows, d.no, d.max_candidates, d.max_det = B, 64, 8, cap, 300
    rc = lib.y5_nms_batched_noalloc(C.byref(d), _lib.current_stream())
    assert rc == _lib.EWORKSPACE and b'scratch' in lib.y5_last_error()
