"""The statistics block of the reference's evaluation loop (test.py:186-258, 278-282) behind the same data:
prediction<->label matching for a whole batch on the GPU (one HIP launch, no host round trip), then the
reference's numpy AP computation on the host, as test.py does it.

    det, count = yolo_amd.nms.nms_padded(inf_out, conf_thres, iou_thres)          # test.py:182
    correct = match_predictions(det, count, targets, height, width, iouv)          # test.py:199, 220-247
    stats.append(batch_stats(det, count, correct, targets))                        # test.py:250
    ...
    p, r, ap, f1, ap_class = ap_per_class(*[np.concatenate(x, 0) for x in zip(*stats)])
"""
import numpy as np
import torch

from yolo_amd import _lib


def match_predictions(det, count, targets, height, width, iouv, check_overflow=False):
    """det (B, max_det, 6) fp32 CUDA [xyxy, conf, cls] and count (B,) int32 from nms_padded; targets (T, 6) fp32
    [image, class, x, y, w, h normalised]; iouv (niou,) fp32.  Clips det to (height, width) in place (test.py:199) and
    returns correct (B, max_det, niou) bool on the device."""
    if det.dtype is not torch.float32 or not det.is_cuda or not det.is_contiguous():
        raise ValueError('match_predictions expects a contiguous fp32 CUDA tensor')
    B, max_det, nf = det.shape
    targets = targets.to(device=det.device, dtype=torch.float32).contiguous()
    iouv = iouv.to(device=det.device, dtype=torch.float32).contiguous()
    correct = torch.empty(B, max_det, iouv.numel(), dtype=torch.uint8, device=det.device)
    overflow = torch.zeros(B, dtype=torch.int32, device=det.device) if check_overflow else None
    with torch.cuda.device(det.device):
        _lib.check(_lib.lib().y5_match_predictions_f32(
            det.data_ptr(), count.data_ptr(), targets.data_ptr() if targets.numel() else None, targets.shape[0], iouv.data_ptr(),
            iouv.numel(), correct.data_ptr(), overflow.data_ptr() if check_overflow else None, B, max_det, nf, float(height),
            float(width), _lib.current_stream()))
    if check_overflow and bool(overflow.any()):
        raise _lib.Y5Error('match_predictions: an image has more labels than the kernel holds (2048)')
    return correct.bool()


def batch_stats(det, count, correct, targets):
    """The tuples test.py:250 appends, one per image: (correct, conf, pcls, tcls) on the host; images without
    predictions and without labels contribute nothing (test.py:196-199)."""
    out = []
    cnt = count.tolist()
    det_c, cor_c, tg = det.cpu(), correct.cpu(), targets.cpu()
    for si, n in enumerate(cnt):
        tcls = tg[tg[:, 0] == si, 1].tolist()
        if n == 0:
            if tcls:
                out.append((torch.zeros(0, correct.shape[2], dtype=torch.bool), torch.Tensor(), torch.Tensor(), tcls))
            continue
        out.append((cor_c[si, :n], det_c[si, :n, 4], det_c[si, :n, 5], tcls))
    return out


def compute_ap(recall, precision):
    """Area under the 101-point interpolated precision envelope (reference: utils/utils.py:278-306, 'interp')."""
    rec = np.concatenate(([0.0], recall, [min(recall[-1] + 1e-3, 1.0)]))
    pre = np.concatenate(([0.0], precision, [0.0]))
    pre = np.maximum.accumulate(pre[::-1])[::-1]        # monotone envelope from the right
    grid = np.linspace(0, 1, 101)
    trapz = getattr(np, 'trapezoid', None) or np.trapz  # numpy < 2.0 only has trapz
    return trapz(np.interp(grid, rec, pre), grid)


def ap_per_class(tp, conf, pred_cls, target_cls, pr_score=0.1):
    """Per-class precision / recall at confidence pr_score, AP per IoU threshold and F1 (reference:
    utils/utils.py:217-275).  tp (n, niou) bool/0-1, conf (n,), pred_cls (n,), target_cls (m,)."""
    order = np.argsort(-conf)
    tp, conf, pred_cls = tp[order], conf[order], pred_cls[order]
    classes = np.unique(target_cls)
    shape = [classes.shape[0], tp.shape[1]]
    ap, p, r = np.zeros(shape), np.zeros(shape), np.zeros(shape)
    for ci, c in enumerate(classes):
        sel = pred_cls == c
        n_gt, n_p = (target_cls == c).sum(), sel.sum()
        if n_p == 0 or n_gt == 0:
            continue
        fpc, tpc = (1 - tp[sel]).cumsum(0), tp[sel].cumsum(0)
        recall = tpc / (n_gt + 1e-16)
        precision = tpc / (tpc + fpc)
        r[ci] = np.interp(-pr_score, -conf[sel], recall[:, 0])
        p[ci] = np.interp(-pr_score, -conf[sel], precision[:, 0])
        for j in range(tp.shape[1]):
            ap[ci, j] = compute_ap(recall[:, j], precision[:, j])
    f1 = 2 * p * r / (p + r + 1e-16)
    return p, r, ap, f1, classes.astype('int32')
