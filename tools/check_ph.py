"""Numerics of the phased convolution kernels (conv_ph.hip) against torch fp32, then their time on the wide layers of
yolov5l / yolov5x at batch 64 beside every other variant.  usage: check_ph.py [check|time|all]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from yolo_amd import _lib

dev = torch.device('cuda:0')
lib = _lib.lib()
NV = lib.y5_conv_num_variants()
NPH = int(os.environ.get('NEW_VARIANTS', str(lib.y5_conv_num_variants_phased())))  # the phased kernels: the last ids
RES = os.environ.get('RES', '0') == '1'           # give every layer a residual input (the Bottleneck shortcut)


def setup(B, Ci, Co, H, W, k, s, seed=0):
    g = torch.Generator().manual_seed(seed)
    cip, cop = (Ci + 7) // 8 * 8, (Co + 7) // 8 * 8
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    x = torch.zeros(B, H, W, cip, dtype=torch.float16)
    x[..., :Ci] = torch.randn(B, H, W, Ci, generator=g).half()
    wt = (torch.randn(Co, k, k, Ci, generator=g) / (Ci * k * k) ** 0.5).half()
    kk = k * k * cip
    kpad, rows = (kk + 63) // 64 * 64, (Co + 255) // 256 * 256
    w = torch.zeros(rows, kpad, dtype=torch.float16)
    wv = torch.zeros(Co, k, k, cip, dtype=torch.float16)
    wv[..., :Ci] = wt
    w[:Co, :kk] = wv.reshape(Co, kk)
    x, w = x.to(dev), w.to(dev)
    y = torch.full((B, Ho, Wo, cop), float('nan'), dtype=torch.float16, device=dev)
    bias = torch.zeros(cop)
    bias[:Co] = torch.randn(Co, generator=g)
    bias = bias.to(dev)
    d = _lib.ConvDesc()
    d.x, d.w, d.y, d.bias = x.data_ptr(), w.data_ptr(), y.data_ptr(), bias.data_ptr()
    d.batch, d.h, d.w_in, d.cin, d.x_ld = B, H, W, cip, cip
    d.ho, d.wo, d.cout, d.y_ld = Ho, Wo, cop, cop
    d.ksize, d.stride, d.pad, d.kpad, d.w_rows = k, s, k // 2, kpad, rows
    d.act, d.slope = _lib.ACT_LEAKY, 0.1
    res = None
    if RES:
        res = torch.zeros(B, Ho, Wo, cop, dtype=torch.float16)
        res[..., :Co] = torch.randn(B, Ho, Wo, Co, generator=g).half()
        res = res.to(dev)
        d.res, d.res_ld = res.data_ptr(), cop
    return d, (x, w, y, bias, wt, res)


def check():
    shapes = [(1, 64, 64, 8, 8, 1, 1), (2, 64, 128, 20, 20, 3, 1), (3, 80, 96, 17, 23, 3, 1), (2, 128, 255, 20, 12, 1, 1),
              (2, 40, 72, 17, 33, 3, 2), (1, 256, 512, 20, 20, 3, 2), (5, 13, 25, 9, 7, 3, 1), (2, 320, 320, 40, 40, 3, 1),
              (3, 128, 128, 80, 80, 3, 1), (7, 64, 200, 33, 47, 3, 1), (2, 160, 96, 19, 20, 3, 1)]
    st = _lib.current_stream()
    for shp in shapes:
        B, Ci, Co, H, W, k, s = shp
        d, (x, w, y, bias, wt, res) = setup(*shp, seed=sum(shp))
        xr = x[..., :Ci].permute(0, 3, 1, 2).float()
        ref = F.leaky_relu((F.conv2d(xr, wt.permute(0, 3, 1, 2).float().to(dev), None, s, k // 2) + bias[:Co].view(1, -1, 1, 1)).half().float(), 0.1).half()
        if RES:
            ref = ref + res[..., :Co].permute(0, 3, 1, 2)
        for v in range(NV - NPH + 1, NV + 1):
            y.fill_(float('nan'))
            d.variant = v
            if lib.y5_conv_f16(C.byref(d), st) == _lib.EINVAL:
                continue
            torch.cuda.synchronize()
            got = y[..., :Co].permute(0, 3, 1, 2)
            bad = torch.isnan(y).any().item()
            err = (got.float() - ref.float()).abs()
            tol = 2e-3 * ref.float().abs() + 2e-3 * ref.float().abs().max()
            print('%-32s v%d  max err %.4g  exact %.4f  %s' % (shp, v, err.max().item(), (err == 0).float().mean().item(),
                                                                'NAN LEFT' if bad else ('ok' if bool((err <= tol).all()) else 'MISMATCH')), flush=True)


def timing():
    shapes = [(64, 128, 128, 80, 80, 3, 1), (64, 256, 256, 40, 40, 3, 1), (64, 512, 512, 20, 20, 3, 1), (64, 256, 512, 40, 40, 3, 2),
              (64, 320, 320, 40, 40, 3, 1), (64, 640, 640, 20, 20, 3, 1), (64, 160, 160, 80, 80, 3, 1), (64, 80, 80, 160, 160, 3, 1),
              (64, 512, 512, 40, 40, 1, 1), (64, 1024, 1024, 20, 20, 1, 1), (64, 256, 256, 40, 40, 1, 1),
              (64, 96, 96, 80, 80, 3, 1), (64, 192, 192, 40, 40, 3, 1), (64, 384, 384, 20, 20, 3, 1), (64, 48, 48, 160, 160, 3, 1),
              (64, 192, 384, 40, 40, 3, 2), (64, 80, 160, 160, 160, 3, 2), (64, 160, 160, 80, 80, 1, 1), (64, 320, 320, 80, 80, 1, 1)]
    st = _lib.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for shp in shapes:
        B, Ci, Co, H, W, k, s = shp
        d, keep = setup(*shp)
        Ho, Wo = d.ho, d.wo
        flops = 2.0 * B * Ho * Wo * Co * k * k * Ci
        res = []
        for v in range(1, NV + 1):
            d.variant = v
            if lib.y5_conv_f16(C.byref(d), st) != 0:
                continue
            lib.y5_conv_f16(C.byref(d), st)
            e0.record()
            for _ in range(10):
                lib.y5_conv_f16(C.byref(d), st)
            e1.record()
            torch.cuda.synchronize()
            res.append((e0.elapsed_time(e1) / 10 * 1e3, v))
        old = min(r for r in res if r[1] <= NV - NPH)
        ph = [r for r in res if r[1] > NV - NPH]
        print('%-34s best other v%-2d %7.1f us %7.1f TF/s | ' % (shp, old[1], old[0], flops / old[0] / 1e6) +
              '  '.join('v%d %7.1f us %7.1f TF/s' % (v, us, flops / us / 1e6) for us, v in ph), flush=True)


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if what in ('check', 'all'):
        check()
    if what in ('time', 'all'):
        timing()
