"""The device half of the reference's image pre-processing: `img.half(); img /= 255.0` (test.py:152-154,
detect.py:57-59) as one HIP kernel.  (The host half -- letterbox, BGR->RGB, HWC->CHW, utils/datasets.py:131-141 --
stays with the data loader.)  A half-precision models.yolo.Model also accepts the uint8 batch directly and folds the
normalisation into its first kernel."""
import torch

from yolo_amd import _lib


def normalize(img, half=True):
    """uint8 tensor (any shape) -> img.half() / 255.0 (half=True, HIP kernel on a GPU tensor) or img.float() / 255.0."""
    if img.dtype is not torch.uint8:
        raise TypeError('normalize expects a uint8 tensor, got %s' % img.dtype)
    if not half or not img.is_cuda:
        return (img.half() if half else img.float()) / 255.0
    img = img.contiguous()
    out = torch.empty(img.shape, dtype=torch.float16, device=img.device)
    with torch.cuda.device(img.device):
        _lib.check(_lib.lib().y5_normalize_u8_f16(img.data_ptr(), out.data_ptr(), img.numel(), _lib.current_stream()))
    return out
