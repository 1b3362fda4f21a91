"""Time the pre-processing step and the stem with and without it folded in (batch 32, 640x640, yolov5s stem)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yolo_amd import _lib, preprocess

dev = torch.device('cuda:0')
lib = _lib.lib()
B, H, W, Co = 32, 640, 640, 32
img8 = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, device=dev)
img16 = preprocess.normalize(img8)
wp = torch.zeros(128, 192, dtype=torch.float16, device=dev)
wp[:Co, :144] = (torch.randn(Co, 144, device=dev) / 10).half()
wp.view(128, 12, 16)[:, :, 12:] = 0
bias = torch.randn(Co, device=dev)
y = torch.zeros(B, H // 2, W // 2, Co, dtype=torch.float16, device=dev)


def desc(img):
    d = _lib.ConvDesc()
    d.x, d.w, d.y, d.bias = img.data_ptr(), wp.data_ptr(), y.data_ptr(), bias.data_ptr()
    d.batch, d.h, d.w_in, d.cin = B, H, W, 3
    d.ho, d.wo, d.cout, d.y_ld = H // 2, W // 2, Co, Co
    d.ksize, d.stride, d.pad, d.kpad, d.w_rows = 3, 1, 1, 192, 128
    d.act, d.slope = _lib.ACT_LEAKY, 0.1
    return d


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


st = _lib.current_stream()
d16, d8 = desc(img16), desc(img8)
t_norm = timeit(lambda: preprocess.normalize(img8))
t_torch = timeit(lambda: img8.half() / 255.0)
t_f16 = timeit(lambda: lib.y5_focus_conv_f16(C.byref(d16), st))
t_u8 = timeit(lambda: lib.y5_focus_conv_u8(C.byref(d8), st))
nb = img8.numel()
print('normalize u8->f16/255: %.1f us (%.0f GB/s read+write); torch half()/255: %.1f us' % (t_norm, 3 * nb / t_norm / 1e3, t_torch))
print('focus+conv from fp16: %.1f us; from uint8 with the normalisation folded in: %.1f us; separate total %.1f us' % (t_f16, t_u8, t_norm + t_f16))
