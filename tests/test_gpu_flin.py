"""The few-row Linear kernel (csrc/i8ie_flin.hip) against the oracle, through the C-ABI: INT32 accumulators and
u8 outputs, every element.  Shapes: AlexNet fc6 / fc7 at the 8-GPU shard size (125 rows) and at a full row group
(128), a single row, 3 rows, K = 1024 (the fewest chunks), K whose chunk count is not a multiple of the stage
count, N not a multiple of 16 / of 4, with and without the fused ReLU; both block shapes (128 rows x 16 features up to 64
rows, 64 x 32 up to 256 rows incl. the 250-row fc6 of a 4-GPU shard); the tiled split-K kernel (variant 11) must
give the same bytes.  The profile hooks confirm which kernel ran."""
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


def _kernels_run(gpu, fn):
    lib = abi.lib()
    abi.ck(lib.i8ie_profile_start(gpu.h, 0))
    try:
        res = fn()
    finally:
        ents = (_Entry * 64)()
        n = C.c_int(0)
        abi.ck(lib.i8ie_profile_stop(gpu.h, ents, 64, C.byref(n)))
    return res, [ents[i].name.decode().split("|")[0] for i in range(n.value)]


SHAPES = [(125, 9216, 4096), (125, 4096, 4096), (128, 4096, 4096), (1, 1024, 256), (100, 1280, 272), (17, 2304, 260),
          (64, 2048, 301), (3, 4096, 4096),
          # round 3, the 64-row x 32-feature form (65 .. 256 rows; above 128 rows three LDS stages, two blocks per CU)
          (65, 2048, 304), (129, 4096, 4096), (250, 9216, 4096), (256, 1280, 272), (200, 1024, 33)]


def _kernel_of(m):
    return "flin_128x16" if m <= 64 else "flin_64x32"


@pytest.mark.parametrize("mkn", SHAPES)
@pytest.mark.parametrize("relu", [False, True])
def test_flin_bit_exact(gpu, orc, mkn, relu):
    m, k, n = mkn
    c = synth.linear_case(orc, 700 + m + k + n, m, k, n)

    def run(variant):
        lib = abi.lib()
        abi.ck(lib.i8ie_ctx_set_option(gpu.h, 2, variant))
        try:
            return gpu.layer_forward_fused("linear", c["q_in"], c["qw"], c["qb"], c["s_in"], c["zp_in"], c["s_w"],
                                           c["s_out"], c["zp_out"], relu=relu)
        finally:
            abi.ck(lib.i8ie_ctx_set_option(gpu.h, 2, 0))

    (out, acc, _), names = _kernels_run(gpu, lambda: run(0 if n >= 2048 else 80))  # (80: below the automatic feature threshold)
    assert _kernel_of(m) in names, names
    want = orc.relu(c["out"], c["zp_out"]) if relu else c["out"]
    assert np.array_equal(acc, c["acc"]) and np.array_equal(out, want)
    (out2, acc2, _), names2 = _kernels_run(gpu, lambda: run(11))
    assert not any(nm.startswith("flin") for nm in names2), names2
    assert np.array_equal(out2, out) and np.array_equal(acc2, acc)
    if 64 < m <= 128:  # the 128-row form is still there for these row counts (variant 81)
        (out3, acc3, _), names3 = _kernels_run(gpu, lambda: run(81))
        assert "flin_128x16" in names3, names3
        assert np.array_equal(out3, out) and np.array_equal(acc3, acc)


def test_flin_extreme_values(gpu, orc):
    """All-255 activations against -128 / +127 weights: the largest accumulators K = 9216 can produce."""
    rng = np.random.default_rng(9)
    n = 256
    q_in = np.full((5, 9216), 255, np.uint8)
    qw = np.full((n, 9216), -128, np.int8)
    qw[1::3] = 127
    qw[2::3] = rng.integers(-128, 128, (len(range(2, n, 3)), 9216))
    qb = rng.integers(-128, 128, n).astype(np.int8)
    for zp_in in (0, 255):
        def run():
            lib = abi.lib()
            abi.ck(lib.i8ie_ctx_set_option(gpu.h, 2, 80))
            try:
                return gpu.layer_forward_fused("linear", q_in, qw, qb, 0.02, zp_in, 0.001, 0.7, 128, relu=False)
            finally:
                abi.ck(lib.i8ie_ctx_set_option(gpu.h, 2, 0))

        (out, acc, _), names = _kernels_run(gpu, run)
        assert "flin_128x16" in names, names
        want, pre, _ = orc.linear(q_in, qw, qb, 0.02, zp_in, 0.001, 0.7, 128, want_acc=True)
        assert np.array_equal(acc, pre) and np.array_equal(out, want)
