"""i8ie -- the reference engine's Python surface, backed by MI355X HIP kernels.

Mirrors reference i8ie/__init__.py:1-32: `tensor`, `argmax`, `relu`,
`max_pool2d`, `quantize`, `dequantize`, `Linear`, `Conv2d`, `Tensor`, `Module`.
Every op forwards to the rebuilt extension `_CXX_i8ie`; tensors are
device-resident and `.numpy()` copies back to the host.
"""
import _CXX_i8ie as _C

from .layer import Conv2d, Layer, Linear
from .module import Module
from .tensor import Tensor

FullyConnected = Linear  # BASELINE.json's name for the same class (no such symbol in the reference)

__all__ = [
    "tensor", "argmax", "relu", "max_pool2d", "quantize", "dequantize",
    "Linear", "FullyConnected", "Conv2d", "Tensor", "Module",
    "synchronize", "set_device", "pinned_empty", "from_torch",
]


def tensor(ndarray):
    """Copy an ndarray (cast to float32) into a new Tensor (reference i8ie/__init__.py:13-14)."""
    return Tensor(_C.tensor(ndarray))


def argmax(x, *args, **kwargs):
    """numpy argmax on the host copy, wrapped again as a float Tensor (reference :17-18)."""
    return tensor(x.numpy().argmax(*args, **kwargs))


def relu(x):
    return Tensor(_C.relu(x.data))


def max_pool2d(x, kernel_size, stride):
    return Tensor(_C.max_pool2d(x.data, kernel_size, stride))


def quantize(x, scale, zero_point):
    return Tensor(_C.quantize(x.data, scale, zero_point))


def dequantize(x):
    return Tensor(_C.dequantize(x.data))


# ---- additive helpers (not in the reference) ---------------------------------
def synchronize():
    """Wait for all queued device work (ops are asynchronous on one HIP stream)."""
    _C.synchronize()


def pinned_empty(shape):
    """float32 ndarray in pinned host memory.  `tensor()` of it (or of a contiguous slice) uploads on
    the transfer stream, beside the kernels of the batch before; keep the contents unchanged until
    `Tensor.wait_upload()` returns."""
    return _C.pinned_empty([int(d) for d in shape])


def from_torch(t, synchronize=True):
    """Additive: wrap a contiguous float32 torch tensor on this GPU as a Tensor without copying (the reference's
    `tensor()` copies an ndarray).  `synchronize` waits for torch's current stream first; pass False when the
    engine was put on that stream with `_CXX_i8ie.use_stream`."""
    import torch

    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise TypeError("from_torch: need a contiguous float32 CUDA tensor")
    if synchronize:
        torch.cuda.current_stream(t.device).synchronize()
    return Tensor(_C.tensor_from_device(t.data_ptr(), [int(d) for d in t.shape], t))


def set_device(index):
    """Choose the GPU before the first op (default: $I8IE_DEVICE, else $LOCAL_RANK, else 0)."""
    _C.set_device(int(index))
