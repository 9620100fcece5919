"""Layer wrappers (reference i8ie/layer.py:5-35)."""
import _CXX_i8ie as _C

from .tensor import Tensor


class Layer:
    """Common behaviour of Linear and Conv2d; `self.layer` is the extension object."""

    layer = None

    def __call__(self, x):
        # FP32 tensor -> FP32 path; uint8 tensor -> INT8 path (after convert())
        return Tensor(self.layer(x.data))

    def load_weight(self, weight):
        self.layer.load_weight(weight)

    def load_bias(self, bias):
        self.layer.load_bias(bias)

    def prepare(self):
        """Start collecting output samples for calibration (reference src/layer.cc:28-35)."""
        self.layer.prepare()

    def convert(self):
        """Quantise weights, fix the output (scale, zero_point) (reference src/layer.cc:36-54)."""
        self.layer.convert()

    # ---- additive (the reference cannot inject or read these) ----------------
    def set_output_qparams(self, scale, zero_point):
        """Use a given output (scale, zero_point) instead of the randomised calibrator's."""
        self.layer.set_output_qparams(float(scale), int(zero_point))

    def output_qparams(self):
        return self.layer.output_qparams()

    def forward_debug(self, x):
        """INT8 forward that also returns the INT32 pre-requant accumulators (numpy)."""
        out, acc = self.layer.forward_debug(x.data)
        return Tensor(out), acc


class Linear(Layer):
    def __init__(self, in_channels, out_channels):
        self.layer = _C.Linear(in_channels, out_channels)


class Conv2d(Layer):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        self.layer = _C.Conv2d(in_channels, out_channels, kernel_size, stride, padding)
