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

    # ---- additive: save / restore the converted (INT8) model ------------------
    # The reference can only load FP32 state dicts (module.py:10-16) and loses the
    # quantised state when the process ends; these keep it as plain arrays.
    def quantized_state_dict(self):
        """{'<attr>.q_weight' int8, '<attr>.q_bias' int8, '<attr>.qparams' float64[3] =
        (weight_scale, out_scale, out_zero_point)} for every converted layer."""
        import numpy as np

        out = {}
        for name, layer in self._layers():
            L = layer.layer
            if not L.is_quantized():
                raise RuntimeError("layer %r is not converted" % name)
            s_out, zp_out = L.output_qparams()
            out[name + ".q_weight"] = L.q_weight()
            out[name + ".q_bias"] = L.q_bias()
            out[name + ".qparams"] = np.array([L.weight_scale(), s_out, zp_out], np.float64)
        return out

    def load_quantized(self, state):
        """Inverse of quantized_state_dict(); marks the module as quantised."""
        import numpy as np

        for name, layer in self._layers():
            w_scale, s_out, zp_out = (float(v) for v in state[name + ".qparams"])
            layer.layer.load_quantized(np.asarray(state[name + ".q_weight"], np.int8),
                                       np.asarray(state[name + ".q_bias"], np.int8),
                                       float(np.float32(w_scale)), float(np.float32(s_out)), int(zp_out))
        self.is_quant = True

    def save_quantized(self, path):
        """Write quantized_state_dict() as an .npz (no pickle)."""
        import numpy as np

        np.savez(path, **self.quantized_state_dict())

    def load_quantized_file(self, path):
        import numpy as np

        with np.load(path, allow_pickle=False) as f:
            self.load_quantized({k: f[k] for k in f.files})
