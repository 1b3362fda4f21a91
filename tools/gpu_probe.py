"""Exploratory GPU check: prints error statistics of every HIP stage against the CPU oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
import util
from yolo_amd import engine, nms as hip_nms, _lib
from oracle import y5ref, nms_c

dev = torch.device('cuda:0')
torch.set_grad_enabled(False)
print('arch', torch.cuda.get_device_properties(0).gcnArchName, flush=True)

def stats(name, got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    d = (got - ref).abs()
    rel = d / (ref.abs() + 1e-3 * ref.abs().max())
    print('%-28s max|d| %.3e  max|ref| %.3e  max rel %.3e  mean|d| %.3e  exact %.4f' % (name, d.max(), ref.abs().max(), rel.max(), d.mean(), (d == 0).float().mean()), flush=True)

ops = y5ref._Ops('half')
for name in util.BLOCKS:
    m, x, y = util.block_from_fixture(name)
    sd = {'model.0.' + k: v for k, v in m.state_dict().items()}
    xh = x.half()
    if name.startswith('conv'):
        ref = y5ref.conv_block(ops, sd, 'model.0', xh.float(), 2 if 's2' in name else 1)
    elif name == 'focus':
        ref = y5ref.conv_block(ops, sd, 'model.0.conv', y5ref.focus_gather(xh.float()))
    elif name.startswith('bottleneck'):
        ref = y5ref.bottleneck(ops, sd, 'model.0', xh.float(), 'noadd' not in name)
    elif name.startswith('csp'):
        n = len(m.m)
        ref = y5ref.bottleneck_csp(ops, sd, 'model.0', xh.float(), n, 'noshortcut' not in name)
    else:
        ref = y5ref.spp(ops, sd, 'model.0', xh.float())
    got = engine.run_block(m.half().to(dev), xh.to(dev))
    stats(name, got, ref)

# decode
lib = _lib.lib()
import ctypes as C
for tag in ('nc80_64x96', 'nc5_128x128', 'nc1_odd'):
    z = util.golden('decode_' + tag)
    zg = torch.from_numpy(z['z'])
    B, rows, no = zg.shape
    na = 3
    zo = torch.zeros(B, rows, no, dtype=torch.float16, device=dev)
    row0 = 0
    ok_x = True
    for i in range(3):
        raw = torch.from_numpy(z['raw%d' % i])  # (B, na*no, ny, nx)
        _, c, ny, nx = raw.shape
        ld = (c + 7) // 8 * 8
        head = torch.zeros(B, ny, nx, ld, dtype=torch.float16, device=dev)
        head[..., :c] = raw.permute(0, 2, 3, 1).to(dev)
        xo = torch.zeros(B, na, ny, nx, no, dtype=torch.float16, device=dev)
        d = _lib.DecodeDesc()
        d.head, d.head_ld, d.x_out, d.z_out = head.data_ptr(), ld, xo.data_ptr(), zo.data_ptr()
        d.batch, d.ny, d.nx, d.na, d.no = B, ny, nx, na, no
        d.z_rows_total, d.z_row0 = rows, row0
        d.stride = float(z['stride'][i])
        ag = z['anchor_grid'][i].reshape(na, 2)
        for a in range(na):
            d.anchor_wh[2 * a], d.anchor_wh[2 * a + 1] = float(ag[a, 0]), float(ag[a, 1])
        _lib.check(lib.y5_detect_decode_f16(C.byref(d), None))
        torch.cuda.synchronize()
        ok_x &= bool((xo.cpu() == torch.from_numpy(z['x%d' % i])).all())
        row0 += na * ny * nx
    a16, b16 = zo.cpu().view(torch.int16).int(), zg.view(torch.int16).int()
    print('decode %-14s x exact %s  z exact frac %.6f  max ulp %d' % (tag, ok_x, (a16 == b16).float().mean(), (a16 - b16).abs().max()), flush=True)

# nms
for name in ['multilabel', 'conf01', 'ties', 'agnostic', 'classes', 'over300', 'nc1', 'empty', 'half_input', 'single']:
    z = util.golden('nms_' + name)
    pred = torch.from_numpy(z['pred']).to(dev)
    classes = list(z['classes']) if len(z['classes']) else None
    det, cnt, kept, ncand = hip_nms.nms_padded(pred, float(z['conf']), float(z['iou']), classes=classes, agnostic=bool(z['agnostic']), return_index=True)
    torch.cuda.synchronize()
    ok_cnt = np.array_equal(cnt.cpu().numpy(), z['count'])
    ok_det = np.array_equal(det.cpu().numpy(), z['det'])
    ref = nms_c.non_max_suppression(z['pred'], float(z['conf']), float(z['iou']), classes=classes, agnostic=bool(z['agnostic']))
    ok_idx = all(np.array_equal(kept[b, :len(r[1])].cpu().numpy(), r[1]) for b, r in enumerate(ref))
    print('nms %-12s count %s det %s idx %s ncand %s' % (name, ok_cnt, ok_det, ok_idx, ncand.tolist()), flush=True)

# boxes nms primitive
g = torch.Generator().manual_seed(0)
bx = torch.rand(3000, 2, generator=g) * 300
boxes = torch.cat((bx, bx + torch.rand(3000, 2, generator=g) * 60 + 5), 1)
sc = torch.rand(3000, generator=g)
k1 = hip_nms.nms(boxes.to(dev), sc.to(dev), 0.5, max_keep=1024).cpu().numpy()
k2 = nms_c.nms(boxes.numpy(), sc.numpy(), 0.5)[:1024]
print('nms boxes primitive', np.array_equal(k1, k2), len(k1), len(k2), flush=True)

# end to end small
for cfg in ('yolov5s', 'yolov5s_ori'):
    model = util.build_model(cfg)
    util.randomize_bn(model, 11)
    model.eval()
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(5))
    sd = model.state_dict()
    import yaml
    cfgd = yaml.safe_load(open(os.path.join(util.CFG_DIR, cfg + '.yaml')))
    mh = model.half().to(dev)
    sdh = {k: v.float().cpu() for k, v in mh.state_dict().items()}
    (zr, xr), feats = y5ref.forward(cfgd, sdh, x.half().float(), precision='half', strides=model.stride, return_features=True)
    zg, xg = mh(x.half().to(dev))
    print(cfg, 'engine calls', __import__('yolo_amd.engine', fromlist=['x']).calls(mh))
    for i in range(3):
        stats('%s x%d' % (cfg, i), xg[i], xr[i])
    stats(cfg + ' z', zg, zr)
    with engine.disabled():
        zt, xt = mh(x.half().to(dev))
    stats(cfg + ' z torch-gpu vs oracle', zt, zr)
    stats(cfg + ' z hip vs torch-gpu', zg, zt)
print('PROBE DONE')
