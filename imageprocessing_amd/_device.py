"""Device-buffer plumbing: PyTorch-ROCm owns every device allocation.

Images live on the GPU as *pitched* uint8 tensors: rows padded to a multiple of
256 bytes so that every row start is 16-byte aligned for the kernels' vector
accesses whatever the image width is.  numpy arrays are copied in (and results
copied out) with strided tensor copies; ROCm tensors that are already suitably
aligned are used in place.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _native


class error(Exception):
    """Raised where ``cv2`` would raise ``cv2.error`` (bad depth, channels, sizes ...)."""


def _device():
    if not torch.cuda.is_available():
        raise error("imageprocessing_amd needs a ROCm GPU: torch.cuda.is_available() is False (there is no CPU path)")
    return torch.device("cuda", torch.cuda.current_device())


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def pitch_for(row_bytes: int) -> int:
    return (max(int(row_bytes), 1) + 255) // 256 * 256


class DevImage:
    """A batch of pitched 8-bit images on the device: storage uint8[batch, height, pitch]."""

    __slots__ = ("t", "batch", "height", "width", "channels")

    def __init__(self, t: torch.Tensor, batch: int, height: int, width: int, channels: int):
        self.t, self.batch, self.height, self.width, self.channels = t, batch, height, width, channels

    @property
    def ptr(self) -> int:
        return self.t.data_ptr()

    @property
    def pitch(self) -> int:
        return self.t.stride(1) if self.t.dim() == 3 else self.t.stride(0)

    @property
    def stride(self) -> int:
        return self.t.stride(0) if self.t.dim() == 3 else self.t.stride(0) * self.height

    @property
    def row_bytes(self) -> int:
        return self.width * self.channels

    @staticmethod
    def empty(batch: int, height: int, width: int, channels: int = 1) -> "DevImage":
        pitch = pitch_for(width * channels)
        t = torch.empty((max(batch, 1), max(height, 1), pitch), dtype=torch.uint8, device=_device())
        return DevImage(t, batch, height, width, channels)

    def valid(self) -> torch.Tensor:
        """uint8[batch, height, width*channels] view without the row padding."""
        return self.t[: self.batch, : self.height, : self.row_bytes]

    def to_numpy(self, squeeze_batch: bool) -> np.ndarray:
        a = self.valid().cpu().numpy()
        if self.channels > 1:
            a = a.reshape(self.batch, self.height, self.width, self.channels)
        return a[0] if squeeze_batch else a

    def to_tensor(self, squeeze_batch: bool) -> torch.Tensor:
        a = self.valid()
        if self.channels > 1:
            a = a.reshape(self.batch, self.height, self.width, self.channels)
        else:
            a = a.contiguous() if self.pitch != self.width else a
        return a[0] if squeeze_batch else a


def upload(arr, channels: int, *, batched: bool = False) -> DevImage:
    """numpy / torch image(s) -> DevImage.  ``channels`` is what the op requires (1 or 3)."""
    is_torch = isinstance(arr, torch.Tensor)
    if not is_torch:
        arr = np.asarray(arr)
        if arr.dtype != np.uint8:
            raise error(f"unsupported depth {arr.dtype}: only 8-bit unsigned (CV_8U) images are supported")
    elif arr.dtype != torch.uint8:
        raise error(f"unsupported depth {arr.dtype}: only 8-bit unsigned (CV_8U) images are supported")
    nd = arr.ndim if not is_torch else arr.dim()
    want = (2 if channels == 1 else 3) + (1 if batched else 0)
    if channels == 1 and nd == want + 1 and arr.shape[-1] == 1:
        arr = arr[..., 0]
        nd -= 1
    if nd != want or (channels == 3 and arr.shape[-1] != 3):
        raise error(f"expected a {'batch of ' if batched else ''}{channels}-channel 8-bit image, got shape {tuple(arr.shape)}")
    if not batched:
        arr = arr[None]
    batch, height, width = int(arr.shape[0]), int(arr.shape[1]), int(arr.shape[2])
    rb = width * channels
    if is_torch and arr.is_cuda:
        flat = arr.reshape(batch, height, rb) if arr.is_contiguous() else arr.contiguous().reshape(batch, height, rb)
        if rb % 16 == 0 and flat.data_ptr() % 16 == 0 and batch * height * rb > 0:
            return DevImage(flat, batch, height, width, channels)  # zero copy
        img = DevImage.empty(batch, height, width, channels)
        if batch * height * rb:
            img.valid().copy_(flat)
        return img
    img = DevImage.empty(batch, height, width, channels)
    if batch * height * rb:
        if is_torch:
            host = arr.contiguous().reshape(batch, height, rb)
        else:
            a = np.ascontiguousarray(arr)
            if any(st < 0 for st in a.strides):  # a reversed axis of length 1 still counts as contiguous
                a = a.copy()
            host = torch.from_numpy(a.reshape(batch, height, rb))
        img.valid().copy_(host)
    return img


def workspace(nbytes: int) -> torch.Tensor:
    return torch.empty((max(int(nbytes), 16) + 255) // 256 * 256, dtype=torch.uint8, device=_device())


def check(rc: int) -> None:
    if rc != 0:
        raise error(f"{_native.last_error()} (code {rc})")
