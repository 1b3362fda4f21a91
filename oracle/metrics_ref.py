"""TEST INFRASTRUCTURE ONLY (imported by tests/, never by the product path).

CPU restatement of the reference's statistics block: the per-image loop of test.py:186-258 with torch CPU ops,
statement by statement.  Its building blocks (box_iou utils/utils.py:355-377, xywh2xyxy :183-190, clip_coords
:209-214) and the AP functions are pinned by fixtures computed with the reference's own functions
(tests/golden/metrics_*.npz, generator tests/golden/make_golden_metrics.py); the loop itself is inline code of
test.test() that cannot be called in isolation, so its order of operations is pinned only through a fixture whose loop
was written in the generator around the reference's functions ("parity unpinned" for the control flow)."""
import torch


def box_iou(box1, box2):
    def area(b):
        return (b[2] - b[0]) * (b[3] - b[1])
    a1, a2 = area(box1.t()), area(box2.t())
    inter = (torch.min(box1[:, None, 2:], box2[:, 2:]) - torch.max(box1[:, None, :2], box2[:, :2])).clamp(0).prod(2)
    return inter / (a1[:, None] + a2 - inter)


def xywh2xyxy(x):
    y = torch.zeros_like(x)
    y[:, 0] = x[:, 0] - x[:, 2] / 2
    y[:, 1] = x[:, 1] - x[:, 3] / 2
    y[:, 2] = x[:, 0] + x[:, 2] / 2
    y[:, 3] = x[:, 1] + x[:, 3] / 2
    return y


def match_image(pred, labels, height, width, iouv):
    """pred (n, 6) fp32 (clipped in place), labels (nl, 5) [class, xywh normalised] -> correct (n, niou) bool."""
    pred[:, 0].clamp_(0, width); pred[:, 1].clamp_(0, height)
    pred[:, 2].clamp_(0, width); pred[:, 3].clamp_(0, height)
    correct = torch.zeros(pred.shape[0], iouv.numel(), dtype=torch.bool)
    nl = labels.shape[0]
    if nl:
        whwh = torch.tensor([width, height, width, height], dtype=torch.float32)
        detected = []
        tcls = labels[:, 0]
        tbox = xywh2xyxy(labels[:, 1:5]) * whwh
        for cls in torch.unique(tcls):
            ti = (cls == tcls).nonzero().view(-1)
            pi = (cls == pred[:, 5]).nonzero().view(-1)
            if pi.shape[0]:
                ious, i = box_iou(pred[pi, :4], tbox[ti]).max(1)
                for j in (ious > iouv[0]).nonzero().view(-1).tolist():
                    d = int(ti[i[j]])
                    if d not in detected:
                        detected.append(d)
                        correct[pi[j]] = ious[j] > iouv
                        if len(detected) == nl:
                            break
    return correct
