"""Generates tests/golden/metrics_*.npz with the reference's own functions (box_iou, xywh2xyxy, clip_coords,
ap_per_class, compute_ap from /root/reference/utils/utils.py), imported as make_golden.py does.  The per-image
matching loop is inline in test.test() (test.py:220-247) and cannot be called: it is written out here around the
reference's functions, on seeded synthetic detections and labels.  numpy 2 removed the alias np.trapz that the
reference's compute_ap calls; it is pointed at np.trapezoid (the same function) for the run.
Run once:  python tests/golden/make_golden_metrics.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def synth(seed, B, n_det, n_lab, nc, size=(384, 640)):
    g = torch.Generator().manual_seed(seed)
    h, w = size
    targets, dets, counts = [], torch.zeros(B, 300, 6), []
    for b in range(B):
        nl = int(torch.randint(0, n_lab + 1, (1,), generator=g)) if b else n_lab
        cls = torch.randint(0, nc, (nl,), generator=g).float()
        cxy = torch.rand(nl, 2, generator=g) * 0.8 + 0.1
        wh = torch.rand(nl, 2, generator=g) * 0.25 + 0.03
        targets.append(torch.cat((torch.full((nl, 1), float(b)), cls[:, None], cxy, wh), 1))
        n = int(torch.randint(0, n_det + 1, (1,), generator=g)) if b else n_det
        # detections: jittered copies of labels (several per label, equal boxes included) plus clutter reaching outside
        boxes = []
        for k in range(n):
            if nl and k % 3 != 2:
                t = int(torch.randint(0, nl, (1,), generator=g))
                jit = (torch.rand(4, generator=g) - 0.5) * (0.0 if k % 7 == 0 else 0.08)
                c = torch.tensor([cxy[t, 0] - wh[t, 0] / 2, cxy[t, 1] - wh[t, 1] / 2, cxy[t, 0] + wh[t, 0] / 2, cxy[t, 1] + wh[t, 1] / 2])
                bx = (c + jit) * torch.tensor([w, h, w, h])
                cl = cls[t] if k % 5 else float(torch.randint(0, nc, (1,), generator=g))
            else:
                p = torch.rand(2, generator=g) * torch.tensor([w * 1.1, h * 1.1]) - 20
                bx = torch.cat((p, p + torch.rand(2, generator=g) * 150 + 4))
                cl = float(torch.randint(0, nc, (1,), generator=g))
            boxes.append(torch.cat((bx, torch.tensor([0.0, float(cl)]))))
        if n:
            d = torch.stack(boxes)
            d[:, 4] = torch.sort(torch.rand(n, generator=g), descending=True).values
            dets[b, :n] = d
        counts.append(n)
    return dets, torch.tensor(counts, dtype=torch.int32), torch.cat(targets, 0), h, w


def main():
    _, _, uu, _ = mg.import_reference()
    np.trapz = np.trapezoid
    iouv = torch.linspace(0.5, 0.95, 10)
    all_stats = []
    for name, args in (('a', (1, 4, 120, 25, 5)), ('b', (2, 6, 300, 60, 80)), ('c', (3, 3, 40, 0, 3))):
        det, count, targets, h, w = synth(*args)
        whwh = torch.Tensor([w, h, w, h])
        clipped = det.clone()
        correct_all = torch.zeros(det.shape[0], 300, 10, dtype=torch.bool)
        for si in range(det.shape[0]):
            n = int(count[si])
            pred = clipped[si, :n]
            labels = targets[targets[:, 0] == si, 1:]
            nl = len(labels)
            tcls = labels[:, 0].tolist() if nl else []
            if n == 0:
                if nl:
                    all_stats.append((torch.zeros(0, 10, dtype=torch.bool), torch.Tensor(), torch.Tensor(), tcls))
                continue
            uu.clip_coords(pred, (h, w))
            correct = torch.zeros(pred.shape[0], 10, dtype=torch.bool)
            if nl:
                detected = []
                tcls_tensor = labels[:, 0]
                tbox = uu.xywh2xyxy(labels[:, 1:5]) * whwh
                for cls in torch.unique(tcls_tensor):
                    ti = (cls == tcls_tensor).nonzero().view(-1)
                    pi = (cls == pred[:, 5]).nonzero().view(-1)
                    if pi.shape[0]:
                        ious, i = uu.box_iou(pred[pi, :4], tbox[ti]).max(1)
                        for j in (ious > iouv[0]).nonzero():
                            d = ti[i[j]]
                            if d not in detected:
                                detected.append(d)
                                correct[pi[j]] = ious[j] > iouv
                                if len(detected) == nl:
                                    break
            correct_all[si, :n] = correct
            all_stats.append((correct, pred[:, 4].clone(), pred[:, 5].clone(), tcls))
        mg.save('metrics_match_' + name, det=det, count=count, targets=targets, height=h, width=w, iouv=iouv,
                clipped=clipped, correct=correct_all)
    stats = [np.concatenate([np.asarray(v) for v in x], 0) for x in zip(*all_stats)]
    p, r, ap, f1, cls = uu.ap_per_class(*stats)
    mg.save('metrics_ap', tp=stats[0], conf=stats[1], pred_cls=stats[2], target_cls=stats[3], p=p, r=r, ap=ap, f1=f1, classes=cls)
    g = torch.Generator().manual_seed(0)
    b1 = torch.rand(40, 2, generator=g) * 300
    b1 = torch.cat((b1, b1 + torch.rand(40, 2, generator=g) * 120), 1)
    b2 = torch.rand(17, 2, generator=g) * 300
    b2 = torch.cat((b2, b2 + torch.rand(17, 2, generator=g) * 120), 1)
    mg.save('metrics_box_iou', box1=b1, box2=b2, iou=uu.box_iou(b1, b2))


if __name__ == '__main__':
    main()
