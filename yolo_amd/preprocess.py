"""The device half of the reference's image pre-processing: `img.half(); img /= 255.0` (test.py:152-154,
detect.py:57-59) as one HIP kernel.  A half-precision models.yolo.Model also accepts the uint8 batch directly and folds
the normalisation into its first kernel - and, through from_hwc, the loader's `img[:, :, ::-1].transpose(2, 0, 1)`
(BGR -> RGB, HWC -> CHW, utils/datasets.py:140-141) as well, so that the letterboxed frames can be uploaded as the decoder
left them.  (Letterboxing itself, utils/datasets.py:131-137, stays with the data loader.)"""
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


def from_hwc(img, bgr=True):
    """A (batch, height, width, 3) uint8 tensor - letterboxed frames as cv2 delivers them - as the (batch, 3, height, width)
    input of models.yolo.Model, without copying: the returned tensor is a permuted view, marked so that the model reverses
    the channel order (bgr=True: the loader's `img[:, :, ::-1]`).  On the HIP engine the first kernel reads the interleaved
    bytes directly; anywhere else the model flips, converts and divides like the reference's statements.  Call it on the
    tensor that is passed to the model (a later .to(device) returns a plain tensor again)."""
    if img.dtype is not torch.uint8 or img.dim() != 4 or img.shape[3] != 3:
        raise TypeError('from_hwc expects a uint8 tensor of shape (batch, height, width, 3)')
    view = img.contiguous().permute(0, 3, 1, 2)
    view.y5_bgr = bool(bgr)
    return view
