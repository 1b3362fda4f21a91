"""Post-processing half of the inference path (mirror of the reference's utils/utils.py, box/NMS part).

`non_max_suppression` keeps the reference signature and return type (utils/utils.py:564-645).
Tensors that live on a GPU are processed by the HIP kernels in yolo_amd/csrc/nms.hip through the
C-ABI library (candidate compaction -> per-image key sort -> greedy bitmask IoU selection); there is
no silent fallback for them: a missing library raises.  Tensors that live in host memory follow the
same stage list with plain torch ops, because the reference API is device-polymorphic and
Model.__init__ / offline tooling run on the host.
"""
import glob
import math
import os
import time

import numpy as np
import torch

MAX_WH = 4096      # class offset in pixels (reference: utils/utils.py:577)
MAX_DET = 300      # detections kept per image (reference: utils/utils.py:578)
TIME_LIMIT = 10.0  # host-path wall clock break (reference: utils/utils.py:579,642)


def make_divisible(x, divisor):
    """Smallest multiple of divisor >= x (reference: utils/utils.py:128-130)."""
    return math.ceil(x / divisor) * divisor


def check_img_size(img_size, s=32):
    """Round the inference size up to a multiple of the largest stride (reference: utils/utils.py:67-72)."""
    new_size = make_divisible(img_size, int(s))
    if new_size != img_size:
        print('WARNING: --img-size %g must be multiple of max stride %g, updating to %g' % (img_size, s, new_size))
    return new_size


def check_file(file):
    """Return `file` if it exists else the first recursive glob hit (reference: utils/utils.py:118-125)."""
    if os.path.isfile(file):
        return file
    hits = glob.glob('./**/' + file, recursive=True)
    assert len(hits), 'File Not Found: %s' % file
    return hits[0]


def check_anchor_order(m):
    """Flip Detect anchors when their area order disagrees with the stride order (reference: utils/utils.py:107-115)."""
    area = m.anchor_grid.prod(-1).view(-1)
    if (area[-1] - area[0]).sign() != (m.stride[-1] - m.stride[0]).sign():
        print('Reversing anchor order')
        m.anchors[:] = m.anchors.flip(0)
        m.anchor_grid[:] = m.anchor_grid.flip(0)


def _like(x):
    return torch.zeros_like(x) if isinstance(x, torch.Tensor) else np.zeros_like(x)


def xyxy2xywh(x):
    """(n,4) corner boxes -> centre/size boxes (reference: utils/utils.py:173-180)."""
    y = _like(x)
    y[:, 0] = (x[:, 0] + x[:, 2]) / 2
    y[:, 1] = (x[:, 1] + x[:, 3]) / 2
    y[:, 2] = x[:, 2] - x[:, 0]
    y[:, 3] = x[:, 3] - x[:, 1]
    return y


def xywh2xyxy(x):
    """(n,4) centre/size boxes -> corner boxes; each half extent is formed separately (reference: utils/utils.py:183-190)."""
    y = _like(x)
    y[:, 0] = x[:, 0] - x[:, 2] / 2
    y[:, 1] = x[:, 1] - x[:, 3] / 2
    y[:, 2] = x[:, 0] + x[:, 2] / 2
    y[:, 3] = x[:, 1] + x[:, 3] / 2
    return y


def clip_coords(boxes, img_shape):
    """Clamp corner boxes into (height, width) in place (reference: utils/utils.py:209-214)."""
    boxes[:, 0].clamp_(0, img_shape[1])
    boxes[:, 1].clamp_(0, img_shape[0])
    boxes[:, 2].clamp_(0, img_shape[1])
    boxes[:, 3].clamp_(0, img_shape[0])


def scale_coords(img1_shape, coords, img0_shape, ratio_pad=None):
    """Map corner boxes from the letterboxed shape back to the original image (reference: utils/utils.py:193-206)."""
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    else:
        gain, pad = ratio_pad[0][0], ratio_pad[1]
    coords[:, [0, 2]] -= pad[0]
    coords[:, [1, 3]] -= pad[1]
    coords[:, :4] /= gain
    clip_coords(coords, img0_shape)
    return coords


def box_iou(box1, box2):
    """Pairwise IoU of (N,4) and (M,4) corner boxes -> (N,M) (reference: utils/utils.py:355-377)."""
    a1 = (box1[:, 2] - box1[:, 0]) * (box1[:, 3] - box1[:, 1])
    a2 = (box2[:, 2] - box2[:, 0]) * (box2[:, 3] - box2[:, 1])
    rb = torch.min(box1[:, None, 2:], box2[:, 2:])
    lt = torch.max(box1[:, None, :2], box2[:, :2])
    inter = (rb - lt).clamp(0).prod(2)
    return inter / (a1[:, None] + a2 - inter)


def _greedy_nms_host(boxes, scores, iou_thres):
    """Greedy NMS on host tensors with the semantics of torchvision.ops.nms (call site utils/utils.py:627):
    stable descending score order, suppress j when inter / (area_i + area_j - inter) > iou_thres.
    Only used for tensors in host memory."""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.int64)
    order = torch.sort(scores, descending=True, stable=True).indices
    b = boxes[order]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    dead = np.zeros(n, dtype=bool)
    keep = []
    for i in range(n):
        if dead[i]:
            continue
        keep.append(i)
        if i + 1 == n:
            break
        w = (torch.min(b[i, 2], b[i + 1:, 2]) - torch.max(b[i, 0], b[i + 1:, 0])).clamp(min=0)
        h = (torch.min(b[i, 3], b[i + 1:, 3]) - torch.max(b[i, 1], b[i + 1:, 1])).clamp(min=0)
        inter = w * h
        ovr = inter / (area[i] + area[i + 1:] - inter)
        dead[i + 1:] |= (ovr.double() > iou_thres).numpy()
    return order[torch.as_tensor(keep, dtype=torch.int64)]


def _merge_boxes(x, i, boxes, scores, iou_thres, redundant=True):
    """Weighted-mean box merge of the kept rows (reference: utils/utils.py:630-639)."""
    iou = box_iou(boxes[i], boxes) > iou_thres
    weights = iou * scores[None]
    x[i, :4] = torch.mm(weights, x[:, :4]).float() / weights.sum(1, keepdim=True)
    if redundant:
        i = i[iou.sum(1) > 1]
    return i


def _nms_stage_list(prediction, conf_thres, iou_thres, merge, classes, agnostic, nms_fn):
    """The reference's per-image stage list (utils/utils.py:570-645) with the box NMS primitive injected."""
    if prediction.dtype is torch.float16:
        prediction = prediction.float()
    nc = prediction[0].shape[1] - 5
    xc = prediction[..., 4] > conf_thres
    multi_label = nc > 1
    t0 = time.time()
    output = [None] * prediction.shape[0]
    for xi, x in enumerate(prediction):
        x = x[xc[xi]]
        if not x.shape[0]:
            continue
        x[:, 5:] *= x[:, 4:5]
        box = xywh2xyxy(x[:, :4])
        if multi_label:
            i, j = (x[:, 5:] > conf_thres).nonzero().t()
            x = torch.cat((box[i], x[i, j + 5, None], j[:, None].float()), 1)
        else:
            conf, j = x[:, 5:].max(1, keepdim=True)
            x = torch.cat((box, conf, j.float()), 1)[conf.view(-1) > conf_thres]
        if classes:
            x = x[(x[:, 5:6] == torch.tensor(classes, device=x.device)).any(1)]
        n = x.shape[0]
        if not n:
            continue
        c = x[:, 5:6] * (0 if agnostic else MAX_WH)
        boxes, scores = x[:, :4] + c, x[:, 4]
        i = nms_fn(boxes, scores, iou_thres)
        if i.shape[0] > MAX_DET:
            i = i[:MAX_DET]
        if merge and (1 < n < 3E3):
            try:
                i = _merge_boxes(x, i, boxes, scores, iou_thres)
            except Exception:  # the reference swallows failures of the merge step (utils/utils.py:637-639)
                pass
        output[xi] = x[i]
        if (time.time() - t0) > TIME_LIMIT:
            break
    return output


def non_max_suppression(prediction, conf_thres=0.1, iou_thres=0.6, merge=False, classes=None, agnostic=False):
    """Confidence filter + per-class-offset batched NMS on Detect output (reference: utils/utils.py:564-645).

    prediction: (B, N, 5+nc) fp16/fp32, rows [cx, cy, w, h, obj, cls...].
    Returns a python list of length B holding None or an (n<=300, 6) fp32 tensor
    [x1, y1, x2, y2, conf, cls] in descending-confidence order, on prediction.device.
    """
    if prediction.is_cuda:
        from yolo_amd import nms as hip_nms
        det, count = hip_nms.nms_padded(prediction, conf_thres, iou_thres, classes=classes, agnostic=agnostic, merge=merge)
        counts = count.tolist()  # the one device->host sync of the call: the list API needs the lengths
        return [det[b, :c].clone() if c > 0 else None for b, c in enumerate(counts)]
    return _nms_stage_list(prediction, conf_thres, iou_thres, merge, classes, agnostic, _greedy_nms_host)
