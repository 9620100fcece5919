"""Tensor wrapper (reference i8ie/tensor.py:4-37)."""
import _CXX_i8ie as _C


class Tensor:
    """Thin handle around an extension tensor (`.data`): float32, uint8 or int8.

    Quantised tensors carry a per-tensor `scale` and `zero_point`
    (reference include/tensor.h:152-154).
    """

    __slots__ = ("data",)

    def __init__(self, data):
        self.data = data

    def numpy(self):
        """Host copy of the contents (device -> host)."""
        return self.data.numpy()

    @property
    def shape(self):
        # from the handle's metadata: no launch, no layout conversion, no device -> host copy
        return tuple(self.data.shape())

    @property
    def scale(self):
        return self.data.scale()

    @property
    def zero_point(self):
        return self.data.zero_point()

    @property
    def dtype(self):
        # the reference's property body is `pass` (i8ie/tensor.py:35-37): always None
        return None

    def reshape(self, *dims):
        """A view sharing the same buffer; one dimension may be -1 (reference :14-15)."""
        return Tensor(self.data.reshape(list(dims)))

    def sum(self):
        return self.numpy().sum()

    def prefetch(self):
        """Additive: upload a host-created tensor to the GPU now instead of at first use."""
        self.data.prefetch()
        return self

    @property
    def __cuda_array_interface__(self):
        """Additive: zero-copy export of a float32 / uint8 tensor (`torch.as_tensor(t, device="cuda")`, CuPy ...).
        Launches pending work, puts the bytes in the reference's NCHW order and waits for the stream, so the
        consumer may use any stream.  The tensor must outlive the consumer's view."""
        name = type(self.data).__name__
        typestr = {"6TensorIfE": "<f4", "6TensorIhE": "|u1", "6TensorIcE": "|i1", "6TensorIiE": "<i4"}.get(name)
        if typestr is None:
            raise TypeError("no array interface for %s" % name)
        ptr = self.data.data_ptr()
        _C.synchronize()
        return {"shape": tuple(self.data.shape()), "typestr": typestr, "data": (ptr, False), "version": 2,
                "strides": None}

    def numpy_async(self):
        """Additive: queue the device -> host copy behind this tensor's kernels and return a future
        (`.result()` -> ndarray, `.done()`), so the next batch can be launched before the wait."""
        return self.data.numpy_async()

    def wait_upload(self):
        """Additive: block until an asynchronous upload from a `pinned_empty` array has finished
        (after this the source array may be refilled)."""
        self.data.wait_upload()
        return self

    def __eq__(self, other):
        # element-wise equality as a float tensor (reference :11-12); NOT wrapped in Tensor there either
        return _C.tensor(self.numpy() == other.numpy())

    __hash__ = None

    def __repr__(self):
        # prints de-quantised values (reference :8-9)
        return repr((self.numpy() - self.zero_point) * self.scale)
