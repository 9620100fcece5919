"""The many-row Linear kernel (tools/diag/csrc/i8ie_lgemm.hip, diagnostic build, variant 82) against the oracle, through the C-ABI: INT32 accumulators (before the
float bias step) and u8 outputs, every element.  Shapes: row counts that are not a multiple of the 64 / 128-row block tile,
both tile heights, odd and even K-tile counts, K shorter than the padded panel, the NHWC-flattened input of fc6 (its own
weight packing), with and without the fused ReLU, extreme operands; the tiled kernel (variant 11) must give the same bytes.
The profile hooks confirm which kernel ran."""
import ctypes as C

import numpy as np
import pytest

import abi
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    c = abi.Ctx(0)
    yield c
    c.close()


class _Entry(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("launches", C.c_uint64), ("total_ms", C.c_double),
                ("total_ops", C.c_double), ("total_bytes", C.c_double)]


def _run(gpu, variant, fn):
    lib = abi.lib()
    abi.ck(lib.i8ie_ctx_set_option(gpu.h, 2, variant))
    abi.ck(lib.i8ie_profile_start(gpu.h, 0))
    try:
        res = fn()
    finally:
        ents = (_Entry * 64)()
        n = C.c_int(0)
        abi.ck(lib.i8ie_profile_stop(gpu.h, ents, 64, C.byref(n)))
        abi.ck(lib.i8ie_ctx_set_option(gpu.h, 2, 0))
    return res, [ents[i].name.decode().split("|")[0] for i in range(n.value)]


SHAPES = [
    # m, k, n, flat (c, h, w) or None
    (1000, 1024, 1024, None),    # 16 x 8 tiles of 64 rows
    (300, 640, 3584, None),      # 5 K tiles (odd), rows 300 = 4 x 64 + 44
    (257, 512, 3328, None),      # the smallest row count it takes; 4 K tiles
    (1000, 2304, 4096, (256, 3, 3)),   # 128-row tiles (8 x 32 blocks), NHWC-flattened input: the permuted panel
    (515, 2048, 2048, None),     # 9 x 16 tiles of 64 rows
    (700, 1040, 1536, None),     # K = 1040: the padded panel (1152) is longer than the activation rows
]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("relu", [False, True])
def test_lgemm_bit_exact(gpu, orc, shape, relu):
    m, k, n, flat = shape
    c = synth.linear_case(orc, 900 + m + k + n, m, k, n)
    want = orc.relu(c["out"], c["zp_out"]) if relu else c["out"]
    for variant, kern in ((82, "lgemm"), (11, "igemm_lin")):
        (out, acc, _), names = _run(gpu, variant, lambda: gpu.layer_forward_fused(
            "linear", c["q_in"], c["qw"], c["qb"], c["s_in"], c["zp_in"], c["s_w"], c["s_out"], c["zp_out"], relu=relu,
            in_nhwc=flat is not None, flat_chw=flat))
        assert any(nm.startswith(kern) for nm in names), names
        assert np.array_equal(acc, c["acc"]), variant
        assert np.array_equal(out, want), variant


def test_lgemm_extreme_operands(gpu, orc):
    """All-255 activations against +127 / -128 weights: the largest accumulators a layer can produce (|C| > 2^24: the
    float bias step of src/fully_connected.cc:44 rounds)."""
    m, k, n = 300, 4096, 3328
    rng = np.random.default_rng(11)
    q_in = np.full((m, k), 255, np.uint8)
    qw = np.where(rng.random((n, k)) < 0.5, 127, -128).astype(np.int8)
    qb = rng.integers(-128, 128, n).astype(np.int8)
    s_in, zp_in, s_w, s_out, zp_out = np.float32(0.02), 3, np.float32(0.004), np.float32(90.0), 131
    want, pre, _ = orc.linear(q_in, qw, qb, s_in, zp_in, s_w, s_out, zp_out, want_acc=True)
    (out, acc, _), names = _run(gpu, 82, lambda: gpu.layer_forward_fused("linear", q_in, qw, qb, s_in, zp_in, s_w, s_out, zp_out))
    assert any(nm.startswith("lgemm") for nm in names), names
    assert np.array_equal(acc, pre) and np.array_equal(out, want)
