"""Host driver of the HIP NMS kernels (yolo_amd/csrc/nms.hip) behind utils.utils.non_max_suppression.

nms_padded() is the allocation-light batched form: fixed-size (B, max_det, 6) detections plus counts,
all on the device and without a host synchronisation; with out= (tensors from nms_buffers) it allocates nothing.
PaddedNms is the same for a serving loop of one shape: outputs, class mask and descriptor are made once and a call
only launches.  nms() is the drop-in for torchvision.ops.boxes.nms on one box set.
"""
import ctypes as C

import torch

from yolo_amd import _lib

MAX_DET = 300               # reference: utils/utils.py:578
DEFAULT_CANDIDATES = 65536  # first-try candidate capacity per image; grows on demand


_PRED_DTYPE = {torch.float16: 0, torch.float32: 1, torch.bfloat16: 2}  # y5_nms_desc.pred_dtype


def _class_mask(classes, nc, device):
    if not classes:
        return None
    mask = torch.zeros(nc, dtype=torch.uint8)
    for c in classes:
        if 0 <= int(c) < nc:
            mask[int(c)] = 1
    return mask.to(device)


def nms_buffers(batch, max_det=MAX_DET, device='cuda'):
    """The four output tensors of nms_padded, for its out= argument: (det (B, max_det, 6) fp32, count (B,) int32,
    kept_index (B, max_det) int32, n_candidates (B,) int32).  Allocate once, pass to every call."""
    dev = torch.device(device)
    return (torch.empty(batch, max_det, 6, dtype=torch.float32, device=dev), torch.empty(batch, dtype=torch.int32, device=dev),
            torch.empty(batch, max_det, dtype=torch.int32, device=dev), torch.empty(batch, dtype=torch.int32, device=dev))


def _check_out(out, B, max_det, dev):
    if len(out) != 4:
        raise ValueError('out must be the four tensors of nms_buffers()')
    want = [((B, max_det, 6), torch.float32), ((B,), torch.int32), ((B, max_det), torch.int32), ((B,), torch.int32)]
    for t, (shape, dtype) in zip(out, want):
        if tuple(t.shape) != shape or t.dtype is not dtype or t.device != dev or not t.is_contiguous():
            raise ValueError('out tensor %s %s does not match %s %s on %s' % (tuple(t.shape), t.dtype, shape, dtype, dev))


def _fits_builtin(lib, B, cap):
    """True when counters and sort keys of (B images, cap candidates each) fit the scratch array inside the library."""
    return lib.y5_nms_workspace_bytes(B, cap) <= _lib.NMS_NOALLOC_BYTES


def nms_padded(prediction, conf_thres=0.1, iou_thres=0.6, classes=None, agnostic=False, max_det=MAX_DET,
               max_candidates=None, return_index=False, check_overflow=True, stream=None, out=None, merge=False):
    """Run the batched pipeline on a (B, N, 5+nc) fp16/fp32 GPU tensor.

    Returns (det, count) or (det, count, kept_index, n_candidates): det (B, max_det, 6) fp32 rows
    [x1, y1, x2, y2, conf, cls] in descending confidence order, count (B,) int32.  With check_overflow
    the per-image candidate totals are read back once and the call is repeated with a larger capacity
    when an image produced more candidates than max_candidates; pass check_overflow=False (and inspect
    n_candidates yourself) to stay asynchronous.

    merge: the reference's merge step after the selection (utils/utils.py:630-639), per image with 1 < n < 3000
    candidates: kept boxes become the score-weighted mean of the candidates overlapping them, kept boxes that no second
    candidate overlaps are dropped (kept_index and count follow).

    out: the tensors of nms_buffers(B, max_det, device), written in place and returned; the call then allocates no
    device memory: counters and sort keys live in a scratch array inside the library (y5_nms_batched_noalloc), which
    all calls on one GPU share, so two such calls must not overlap in time on one GPU (calls on one stream never do).
    A request that does not fit that array (more than batch 32 x 65536 candidates) raises Y5Error; so does an overflow
    with check_overflow, instead of the silent retry (which would have to allocate).  With classes the class mask is
    still a small allocation per call: use PaddedNms, which keeps it.

    stream: a torch.cuda.Stream to run the pipeline on instead of the current one.  The call is ordered after the work
    already issued on the current stream (the forward that produced `prediction`) and, with check_overflow=False,
    returns at once (with check_overflow the host waits for `stream` to read the candidate totals); the current
    stream does NOT wait for it, so a following forward of the next batch runs beside this NMS (its kernels are small
    and leave most of the chip idle).  The results belong to `stream`: make the consumer wait for it
    (torch.cuda.current_stream().wait_stream(stream), or stream.synchronize()) before reading them.
    """
    if not prediction.is_cuda:
        raise _lib.Y5Error('nms_padded needs a GPU tensor')
    if prediction.dtype not in _PRED_DTYPE:
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
    if out is not None:
        _check_out(out, B, max_det, dev)
        if not _fits_builtin(lib, B, cap):
            raise _lib.Y5Error('nms_padded(out=...): batch %d x %d candidates does not fit the built-in scratch array; '
                               'lower max_candidates, split the batch, or call without out=' % (B, cap))
    if stream is not None:
        stream.wait_stream(torch.cuda.current_stream(dev))
        prediction.record_stream(stream)
    while True:
        with torch.cuda.device(dev), torch.cuda.stream(stream):
            d = _lib.NmsDesc()
            if out is None:
                ws_bytes = lib.y5_nms_workspace_bytes(B, cap)
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
                det, count, kept, ncand = nms_buffers(B, max_det, dev)
                d.workspace, d.workspace_bytes = ws.data_ptr(), ws_bytes
            else:
                det, count, kept, ncand = out
            d.pred = prediction.data_ptr()
            d.class_keep = mask.data_ptr() if mask is not None else None
            d.det, d.count, d.kept_index, d.n_candidates = det.data_ptr(), count.data_ptr(), kept.data_ptr(), ncand.data_ptr()
            d.iou_thres, d.conf_thres = float(iou_thres), float(conf_thres)
            d.pred_dtype = _PRED_DTYPE[prediction.dtype]
            d.batch, d.n_rows, d.no = B, N, no
            d.agnostic = (_lib.NMS_AGNOSTIC if agnostic else 0) | (_lib.NMS_MERGE if merge else 0)
            d.max_candidates, d.max_det = cap, max_det
            _lib.check((lib.y5_nms_batched if out is None else lib.y5_nms_batched_noalloc)(C.byref(d), _lib.current_stream()))
            # the candidate totals are read back on the stream that computed them (inside this context the reduction
            # and the copy are ordered after the kernels; on the caller's stream they could run before them)
            worst = int(ncand.max().item()) if check_overflow else 0
        if worst > cap:
            if out is not None:
                raise _lib.Y5Error('nms_padded(out=...): an image produced %d candidates, more than max_candidates=%d' % (worst, cap))
            cap = worst
            continue
        break
    return (det, count, kept, ncand) if return_index else (det, count)


class PaddedNms:
    """nms_padded for a serving loop over one prediction shape: output tensors (a ring of `ring` sets, used in turn), class
    mask, workspace and descriptors are made here, once; a call checks the shape, sets one pointer and launches: no
    device allocation, no host synchronisation.  The results of call k are overwritten by call k + ring.  Candidate
    overflow is not checked per call: read n_candidates (the fourth output) when it matters; rows past max_candidates
    are dropped lowest-confidence first, as in nms_padded(check_overflow=False).

    Counters and sort keys live in the scratch array inside the library when they fit it (y5_nms_batched_noalloc; every
    such call on one GPU shares that array, so they must not overlap in time: keep them on one stream), else in a
    workspace allocated here."""

    def __init__(self, batch, n_rows, no, dtype=torch.float16, device='cuda', conf_thres=0.1, iou_thres=0.6, classes=None,
                 agnostic=False, max_det=MAX_DET, max_candidates=None, ring=1, merge=False, private_workspace=False):
        if dtype not in _PRED_DTYPE:
            raise ValueError('PaddedNms takes fp16, bf16 or fp32 predictions')
        lib = _lib.lib()
        self.dev = torch.device(device)
        if self.dev.index is None:
            self.dev = torch.device('cuda', torch.cuda.current_device())
        self.shape, self.dtype = (batch, n_rows, no), dtype
        nc = no - 5
        cap = int(max_candidates or min(n_rows * nc, DEFAULT_CANDIDATES))
        self.mask = _class_mask(classes, nc, self.dev)
        # private_workspace: never the library's built-in array (which all calls on one GPU share): for instances whose
        # calls overlap in time with other NMS calls on the same GPU (one instance per stream)
        self.builtin = _fits_builtin(lib, batch, cap) and not private_workspace
        self.fn = lib.y5_nms_batched_noalloc if self.builtin else lib.y5_nms_batched
        self.ws = None
        if not self.builtin:
            self.ws = torch.empty(lib.y5_nms_workspace_bytes(batch, cap), dtype=torch.uint8, device=self.dev)
        self.sets = []
        for _ in range(max(1, int(ring))):
            bufs = nms_buffers(batch, max_det, self.dev)
            d = _lib.NmsDesc()
            d.class_keep = self.mask.data_ptr() if self.mask is not None else None
            if self.ws is not None:
                d.workspace, d.workspace_bytes = self.ws.data_ptr(), self.ws.numel()
            d.det, d.count, d.kept_index, d.n_candidates = [t.data_ptr() for t in bufs]
            d.iou_thres, d.conf_thres = float(iou_thres), float(conf_thres)
            d.pred_dtype = _PRED_DTYPE[dtype]
            d.batch, d.n_rows, d.no = batch, n_rows, no
            d.agnostic = (_lib.NMS_AGNOSTIC if agnostic else 0) | (_lib.NMS_MERGE if merge else 0)
            d.max_candidates, d.max_det = cap, max_det
            self.sets.append((bufs, d, C.byref(d), torch.cuda.Event()))
        self.turn = 0
        self.calls = 0

    def __call__(self, prediction, stream=None, after_current=True):
        """(det, count, kept_index, n_candidates) of `prediction`, computed on the current stream, or on `stream` (a
        torch.cuda.Stream) ordered after the work already issued on the current one: the current stream does not wait for
        it, so the next forward runs beside this NMS; the results then belong to `stream`.  after_current=False skips that
        ordering: for a prediction that was itself produced on `stream`."""
        if (tuple(prediction.shape) != self.shape or prediction.dtype is not self.dtype or prediction.device != self.dev
                or not prediction.is_contiguous()):
            raise ValueError('PaddedNms was built for a contiguous %s %s tensor on %s' % (self.shape, self.dtype, self.dev))
        bufs, d, ref, done = self.sets[self.turn]
        self.turn = (self.turn + 1) % len(self.sets)
        self.calls += 1
        d.pred = prediction.data_ptr()
        if stream is None:
            _lib.check(self.fn(ref, _lib.current_stream()))
            done.record()
        else:
            if after_current:
                stream.wait_stream(torch.cuda.current_stream(self.dev))
            _lib.check(self.fn(ref, C.c_void_p(stream.cuda_stream)))
            done.record(stream)
        return bufs

    def wait_oldest(self):
        """Make the current stream wait for the call made `ring` calls ago, the one whose output set the next call reuses.
        A producer that recycles its own buffers with the same period (engine.set_output_ring of the same length) calls
        this before it overwrites the prediction that call was reading on another stream."""
        if self.calls >= len(self.sets):
            torch.cuda.current_stream(self.dev).wait_event(self.sets[self.turn][3])


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
