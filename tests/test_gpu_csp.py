"""BottleneckCSP with one Bottleneck as one kernel (y5_csp_f16) against the oracle and against the layer-by-layer plan of
the same block (tolerances of tests/test_gpu_conv.py: multi-layer blocks)."""
import ctypes as C

import pytest
import torch

import util
from oracle import y5ref
from test_gpu_conv import close
from yolo_amd import _lib, engine
from yolo_amd.models import common
from yolo_amd.utils import torch_utils

pytestmark = pytest.mark.gpu
OPS16 = y5ref._Ops('half')


def _block(shortcut, fused, seed):
    torch.manual_seed(seed)
    m = common.BottleneckCSP(64, 64, n=1, shortcut=shortcut)
    torch_utils.initialize_weights(m)
    util.randomize_bn(m, seed)
    m.eval()
    if fused:
        for c in m.modules():
            if type(c) is common.Conv:
                c.conv = torch_utils.fuse_conv_and_bn(c.conv, c.bn)
                c.bn = None
                c.forward = c.fuseforward
    return m


def _run(monkeypatch, m, x, mode):
    monkeypatch.setenv('Y5_CSP_FUSE', mode)
    plan = engine.Plan(m, x, block=True)
    try:
        y = plan.run_block(x)
        torch.cuda.synchronize()
        return y, list(plan.labels)
    finally:
        plan.close()


@pytest.mark.parametrize('shape', [(1, 64, 8, 32), (2, 64, 16, 32), (1, 64, 8, 64), (3, 64, 40, 96)], ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('fused', [True, False], ids=['fusedbn', 'bn'])
@pytest.mark.parametrize('shortcut', [True, False], ids=['add', 'noadd'])
def test_one_kernel_block_against_oracle_and_layers(dev, monkeypatch, shortcut, fused, shape):
    m = _block(shortcut, fused, seed=sum(shape) + shortcut)
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(3)).half()
    sd = {'model.0.' + k: v for k, v in m.state_dict().items()}
    ref = y5ref.bottleneck_csp(OPS16, sd, 'model.0', x.float(), 1, shortcut)
    mh = m.half().to(dev)
    got, labels = _run(monkeypatch, mh, x.to(dev), 'force')
    assert any('one kernel' in l and 'whole block' in l for l in labels), labels
    lay, labels2 = _run(monkeypatch, mh, x.to(dev), '0')
    assert not any('whole block' in l for l in labels2)
    assert got.shape == ref.shape
    assert close(got, ref) > 0.98
    assert close(got, lay) > 0.98


def test_one_kernel_block_at_layer_size(dev, monkeypatch):
    """yolov5s layer 2 at 640x640 input: 64 channels at 160x160; against the layer-by-layer plan, and image slots are
    independent (a rolled batch gives the rolled result bit for bit)."""
    m = _block(True, True, seed=5).half().to(dev)
    x = torch.randn(5, 64, 160, 160, generator=torch.Generator().manual_seed(4)).half().to(dev)
    got, _ = _run(monkeypatch, m, x, 'force')
    lay, _ = _run(monkeypatch, m, x, '0')
    assert close(got, lay) > 0.98
    got2, _ = _run(monkeypatch, m, torch.roll(x, 2, 0), 'force')
    assert torch.equal(got2, torch.roll(got, 2, 0))


def test_shapes_outside_the_compiled_plan_are_refused(dev):
    lib = _lib.lib()
    d = _lib.CspDesc()
    buf = torch.zeros(8 * 32 * 64, dtype=torch.float16, device=dev)
    d.x, d.y, d.batch, d.h, d.w = buf.data_ptr(), buf.data_ptr(), 1, 8, 32
    d.c1, d.x_ld, d.ch, d.c2, d.y_ld, d.slope = 64, 64, 16, 64, 64, 0.1
    assert lib.y5_csp_f16(C.byref(d), _lib.current_stream()) == _lib.EINVAL
    d.ch, d.h = 32, 12
    assert lib.y5_csp_f16(C.byref(d), _lib.current_stream()) == _lib.EINVAL
    d.h = 8
    assert lib.y5_csp_f16(C.byref(d), _lib.current_stream()) == _lib.EINVAL   # no weights
