"""Module base class (reference i8ie/module.py:6-35)."""
import _CXX_i8ie as _C

from .layer import Layer
from .tensor import Tensor

# Hard-coded input quantisation of the reference (i8ie/module.py:20).
INPUT_SCALE = 0.025
INPUT_ZERO_POINT = 127


class Module:
    """Subclass, create layers as attributes in __init__, define forward(x)."""

    def __init__(self):
        self.is_quant = False

    def _layers(self):
        return [(k, v) for k, v in self.__dict__.items() if isinstance(v, Layer)]

    def load(self, state_dict):
        """Load a torch-style state dict with keys '<attr>.weight' / '<attr>.bias' (reference :10-16)."""
        for key in state_dict:
            name, attr = key.split(".")
            if attr == "weight":
                self.__dict__[name].load_weight(state_dict[key])
            elif attr == "bias":
                self.__dict__[name].load_bias(state_dict[key])

    def __call__(self, x):
        if self.is_quant:
            x = Tensor(_C.quantize(x.data, INPUT_SCALE, INPUT_ZERO_POINT))
        x = self.forward(x)
        if self.is_quant:
            x = Tensor(_C.dequantize(x.data))
        return x

    def prepare(self):
        for _, layer in self._layers():
            layer.prepare()

    def convert(self):
        for _, layer in self._layers():
            layer.convert()
        self.is_quant = True
