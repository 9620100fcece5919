"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports
every symbol include/i8ie_hip.h declares; host-only behaviour of the i8ie
package (no compute calls: there is no GPU in the build container)."""
import ctypes as C

import numpy as np
import pytest

import abi


def test_library_exports_every_declared_symbol():
    names = abi.declared_symbols()
    assert len(names) >= 30
    lib = abi.lib()
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_version_and_error_channel():
    lib = abi.lib()
    assert lib.i8ie_version() == 1
    # argument errors are reported without touching a GPU
    rc = lib.i8ie_device_count(None)
    assert rc == -1 and b"null" in lib.i8ie_last_error()
    rc = lib.i8ie_sync(None)
    assert rc == -1


def test_quantize_weight_host_entry_matches_oracle(orc):
    rng = np.random.default_rng(3)
    w = rng.normal(0, 0.05, (20, 10, 3, 3)).astype(np.float32)
    b = rng.normal(0, 0.05, 20).astype(np.float32)
    qw = np.empty(w.shape, np.int8)
    qb = np.empty(b.shape, np.int8)
    s = C.c_float()
    rc = abi.lib().i8ie_quantize_weight(w.ctypes.data_as(C.c_void_p), C.c_int64(w.size), b.ctypes.data_as(C.c_void_p),
                                        C.c_int64(b.size), qw.ctypes.data_as(C.c_void_p),
                                        qb.ctypes.data_as(C.c_void_p), C.byref(s))
    assert rc == 0
    oqw, oqb, os_ = orc.quantize_weight(w, b)
    assert np.array_equal(qw, oqw) and np.array_equal(qb, oqb) and np.float32(s.value) == os_


# ---- the Python surface (reference i8ie/*.py, unittest/test_refcount.py, test_tensor_ops.py) ----
@pytest.fixture(scope="module")
def i8ie():
    import int8inferenceengine_amd  # noqa: F401
    import i8ie as mod

    return mod


def test_surface_names(i8ie):
    for name in ["tensor", "argmax", "relu", "max_pool2d", "quantize", "dequantize", "Linear", "Conv2d", "Tensor",
                 "Module"]:
        assert hasattr(i8ie, name)
    import _CXX_i8ie as cx

    for name in ["tensor", "quantize", "dequantize", "relu", "max_pool2d", "Linear", "Conv2d", "6TensorIfE",
                 "6TensorIhE", "6TensorIcE"]:
        assert hasattr(cx, name), name


def test_from_numpy_roundtrip_and_cast(i8ie):
    a = np.random.default_rng(0).uniform(-100, 100, (4, 4)).astype(np.float32)
    t = i8ie.tensor(a)
    assert np.array_equal(t.numpy(), a) and t.numpy().dtype == np.float32
    assert i8ie.tensor(np.arange(6)).numpy().dtype == np.float32  # forcecast (include/tensor.h:40)
    assert t.scale == 1 and t.zero_point == 0 and t.dtype is None
    # non-contiguous input is made contiguous (the reference ignores strides)
    assert np.array_equal(i8ie.tensor(a.T).numpy(), a.T)


def test_reshape_rules(i8ie):
    a = np.arange(16, dtype=np.float32).reshape(4, 4)
    t = i8ie.tensor(a)
    assert np.array_equal(t.reshape(-1, 2).numpy(), a.reshape(-1, 2))
    assert t.reshape(8, -1).shape == (8, 2)
    assert t.reshape(16).shape == (16,)
    for bad in [(-1, -1), (0, 16), (3, -1), (5, 5)]:  # include/tensor.h:114,117,124,130
        with pytest.raises(RuntimeError):
            t.reshape(*bad)


def test_refcount_semantics(i8ie):
    # unittest/test_refcount.py:11-34
    t = i8ie.tensor(np.zeros((4, 4), np.float32))
    t = t
    assert t.data.ref_count() == 1
    b = t
    c = t
    c = 0  # noqa: F841
    assert t.data.ref_count() == 1 and b.data.ref_count() == 1
    t.reshape(-1, 2)
    c = t.reshape(4, -1)
    assert t.data.ref_count() == 2 and c.data.ref_count() == 2
    del c
    assert t.data.ref_count() == 1


def test_sum_argmax_eq(i8ie):
    a = np.arange(16, dtype=np.float32).reshape(-1, 4)
    t = i8ie.tensor(a)
    assert t.sum() == a.sum()
    assert t.data.sum() == 120.0
    assert np.array_equal(i8ie.argmax(t, 0).numpy(), np.argmax(a, 0))
    assert np.array_equal(i8ie.argmax(t, 1).numpy(), np.argmax(a, 1))
    eq = (t == i8ie.tensor(a))  # extension tensor of 0/1 floats (reference i8ie/tensor.py:11-12)
    assert eq.sum() == 16.0
    assert "array" in repr(t)


def test_module_load_and_state_machine(i8ie):
    class Net(i8ie.Module):
        def __init__(self):
            super().__init__()
            self.fc = i8ie.Linear(8, 4)
            self.conv = i8ie.Conv2d(2, 3, 3, padding=1)

        def forward(self, x):
            return self.fc(x)

    net = Net()
    assert net.is_quant is False
    net.load({"fc.weight": np.ones((4, 8), np.float32), "fc.bias": np.zeros(4, np.float32),
              "conv.weight": np.ones((3, 2, 3, 3), np.float32), "conv.bias": np.zeros(3, np.float32),
              "fc.other": 1})
    assert [k for k, _ in net._layers()] == ["fc", "conv"]
    with pytest.raises(RuntimeError):
        i8ie.Conv2d(1, 1, 3, stride=0)  # include/conv2d.h:12-14
    assert net.fc.output_qparams() == (1.0, 0)  # include/layer.h:46-47
    net.fc.set_output_qparams(0.5, 17)
    assert net.fc.output_qparams() == (0.5, 17)
    with pytest.raises(RuntimeError):
        net.fc.layer.q_weight()  # not converted yet


def test_device_ops_fail_loudly_without_gpu(i8ie):
    n = C.c_int(0)
    if abi.lib().i8ie_device_count(C.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    t = i8ie.tensor(np.zeros((2, 2), np.float32))
    with pytest.raises(RuntimeError):
        i8ie.quantize(t, 0.025, 127)  # no CPU fallback
