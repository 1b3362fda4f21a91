"""Host driver of the HIP NMS kernels (yolo_amd/csrc/nms.hip) behind utils.utils.non_max_suppression.

nms_padded() is the allocation-light batched form: fixed-size (B, max_det, 6) detections plus counts,
all on the device and without a host synchronisation; nms() is the drop-in for
torchvision.ops.boxes.nms on one box set.
"""
import ctypes as C

import torch

from yolo_amd import _lib

MAX_DET = 300               # reference: utils/utils.py:578
DEFAULT_CANDIDATES = 65536  # first-try candidate capacity per image; grows on demand


def _class_mask(classes, nc, device):
    if not classes:
        return None
    mask = torch.zeros(nc, dtype=torch.uint8)
    for c in classes:
        if 0 <= int(c) < nc:
            mask[int(c)] = 1
    return mask.to(device)


def nms_padded(prediction, conf_thres=0.1, iou_thres=0.6, classes=None, agnostic=False, max_det=MAX_DET,
               max_candidates=None, return_index=False, check_overflow=True, stream=None):
    """Run the batched pipeline on a (B, N, 5+nc) fp16/fp32 GPU tensor.

    Returns (det, count) or (det, count, kept_index, n_candidates): det (B, max_det, 6) fp32 rows
    [x1, y1, x2, y2, conf, cls] in descending confidence order, count (B,) int32.  With check_overflow
    the per-image candidate totals are read back once and the call is repeated with a larger capacity
    when an image produced more candidates than max_candidates; pass check_overflow=False (and inspect
    n_candidates yourself) to stay asynchronous.

    stream: a torch.cuda.Stream to run the pipeline on instead of the current one.  The call is ordered after the work
    already issued on the current stream (the forward that produced `prediction`) and, with check_overflow=False,
    returns at once (with check_overflow the host waits for `stream` to read the candidate totals); the current
    stream does NOT wait for it, so a following forward of the next batch runs beside this NMS (its kernels are small
    and leave most of the chip idle).  The results belong to `stream`: make the consumer wait for it
    (torch.cuda.current_stream().wait_stream(stream), or stream.synchronize()) before reading them.
    """
    if not prediction.is_cuda:
        raise _lib.Y5Error('nms_padded needs a GPU tensor')
    if prediction.dtype not in (torch.float16, torch.float32):
        prediction = prediction.float()
    prediction = prediction.contiguous()
    B, N, no = prediction.shape
    nc = no - 5
    lib = _lib.lib()
    dev = prediction.device
    if B == 0:
        empty = (torch.zeros(0, max_det, 6, device=dev), torch.zeros(0, dtype=torch.int32, device=dev))
        return empty + ((torch.zeros(0, max_det, dtype=torch.int32, device=dev), torch.zeros(0, dtype=torch.int32, device=dev)) if return_index else ())
    cap = int(max_candidates or min(N * nc, DEFAULT_CANDIDATES))
    mask = _class_mask(classes, nc, dev)
    if stream is not None:
        stream.wait_stream(torch.cuda.current_stream(dev))
        prediction.record_stream(stream)
    while True:
        with torch.cuda.device(dev), torch.cuda.stream(stream):
            ws_bytes = lib.y5_nms_workspace_bytes(B, cap)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            det = torch.empty(B, max_det, 6, dtype=torch.float32, device=dev)
            count = torch.empty(B, dtype=torch.int32, device=dev)
            kept = torch.empty(B, max_det, dtype=torch.int32, device=dev)
            ncand = torch.empty(B, dtype=torch.int32, device=dev)
            d = _lib.NmsDesc()
            d.pred = prediction.data_ptr()
            d.class_keep = mask.data_ptr() if mask is not None else None
            d.workspace, d.workspace_bytes = ws.data_ptr(), ws_bytes
            d.det, d.count, d.kept_index, d.n_candidates = det.data_ptr(), count.data_ptr(), kept.data_ptr(), ncand.data_ptr()
            d.iou_thres, d.conf_thres = float(iou_thres), float(conf_thres)
            d.pred_dtype = 0 if prediction.dtype is torch.float16 else 1
            d.batch, d.n_rows, d.no = B, N, no
            d.agnostic = 1 if agnostic else 0
            d.max_candidates, d.max_det = cap, max_det
            _lib.check(lib.y5_nms_batched(C.byref(d), _lib.current_stream()))
            # the candidate totals are read back on the stream that computed them (inside this context the reduction
            # and the copy are ordered after the kernels; on the caller's stream they could run before them)
            worst = int(ncand.max().item()) if check_overflow else 0
        if worst > cap:
            cap = worst
            continue
        break
    return (det, count, kept, ncand) if return_index else (det, count)


def nms(boxes, scores, iou_thres, max_keep=MAX_DET):
    """Greedy NMS of one box set on the GPU; returns the kept indices (int64) in descending score order,
    truncated to max_keep, which is all non_max_suppression uses (utils/utils.py:627-629)."""
    if not boxes.is_cuda:
        raise _lib.Y5Error('nms needs GPU tensors')
    lib = _lib.lib()
    n = boxes.shape[0]
    boxes = boxes.float().contiguous()
    scores = scores.float().contiguous()
    dev = boxes.device
    with torch.cuda.device(dev):
        keep = torch.empty(max_keep, dtype=torch.int64, device=dev)
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        ws_bytes = max(lib.y5_nms_workspace_bytes(1, max(n, 1)), 512)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        _lib.check(lib.y5_nms_boxes_f32(boxes.data_ptr(), scores.data_ptr(), n, float(iou_thres), keep.data_ptr(),
                                        cnt.data_ptr(), max_keep, ws.data_ptr(), ws_bytes, _lib.current_stream()))
    return keep[:int(cnt.item())]


def scale_detections(det, count, img1_shape, img0_shapes, ratio_pads=None, round_=False):
    """scale_coords + clip_coords (reference: utils/utils.py:193-214) for a whole padded batch, in place on the GPU.
    det (B, max_det, 6) fp32 and count (B,) as returned by nms_padded; img1_shape = (h, w) of the network input;
    img0_shapes[b] = (h, w, ...) of original image b; ratio_pads[b] = ((gain, ...), (pad_x, pad_y)) or None, exactly
    the reference's ratio_pad argument.  round_=True adds detect.py:93's .round().  Returns det."""
    B, max_det, nf = det.shape
    if det.dtype is not torch.float32 or not det.is_cuda or not det.is_contiguous():
        raise ValueError('scale_detections expects a contiguous fp32 CUDA tensor')
    rows = []
    for b in range(B):
        s0 = img0_shapes[b]
        rp = None if ratio_pads is None else ratio_pads[b]
        if rp is None:  # calculate from img0_shape
            gain = min(img1_shape[0] / s0[0], img1_shape[1] / s0[1])
            pad = (img1_shape[1] - s0[1] * gain) / 2, (img1_shape[0] - s0[0] * gain) / 2
        else:
            gain = rp[0][0]
            pad = rp[1]
        rows.append([pad[0], pad[1], gain, s0[1], s0[0]])
    params = torch.tensor(rows, dtype=torch.float64).to(torch.float32).to(det.device)
    with torch.cuda.device(det.device):
        _lib.check(_lib.lib().y5_scale_coords_f32(det.data_ptr(), count.data_ptr(), params.data_ptr(), B, max_det, nf,
                                                   int(bool(round_)), _lib.current_stream()))
    return det
