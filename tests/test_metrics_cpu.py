"""Statistics block (test.py:186-258, ap_per_class / compute_ap): the oracle and the host mirror against fixtures
computed with the reference's own functions (tests/golden/make_golden_metrics.py)."""
import numpy as np
import pytest
import torch

import util
from oracle import metrics_ref
from yolo_amd import metrics


def test_oracle_box_iou_equals_the_reference():
    z = util.golden('metrics_box_iou')
    got = metrics_ref.box_iou(torch.from_numpy(z['box1']), torch.from_numpy(z['box2']))
    assert np.array_equal(got.numpy().view(np.int32), z['iou'].view(np.int32))


@pytest.mark.parametrize('name', ['a', 'b', 'c'])
def test_oracle_matching_equals_the_reference_fixture(name):
    z = util.golden('metrics_match_' + name)
    det, targets = torch.from_numpy(z['det']).clone(), torch.from_numpy(z['targets'])
    iouv = torch.from_numpy(z['iouv'])
    for si in range(det.shape[0]):
        n = int(z['count'][si])
        labels = targets[targets[:, 0] == si, 1:]
        correct = metrics_ref.match_image(det[si, :n], labels, float(z['height']), float(z['width']), iouv)
        assert np.array_equal(correct.numpy(), z['correct'][si, :n])
    assert np.array_equal(det.numpy().view(np.int32), z['clipped'].view(np.int32))


def test_ap_per_class_mirror_equals_the_reference():
    z = util.golden('metrics_ap')
    p, r, ap, f1, classes = metrics.ap_per_class(z['tp'], z['conf'], z['pred_cls'], z['target_cls'])
    assert np.array_equal(classes, z['classes'])
    for got, ref in ((p, z['p']), (r, z['r']), (ap, z['ap']), (f1, z['f1'])):
        assert got.shape == ref.shape and np.allclose(got, ref, rtol=1e-12, atol=0)
    assert float(ap.mean()) > 0.05   # the fixture is not degenerate
