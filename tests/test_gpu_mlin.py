"""The many-row Linear kernel (csrc/i8ie_mlin.hip: 128 rows x 128 features x all of K per block, loader waves feeding a
four-stage LDS ring) against the oracle, through the C-ABI: INT32 accumulators and u8 outputs, every element.  Shapes: AlexNet
fc6 / fc7 at 1000 and 500 rows, row counts that leave the last row tile ragged (257, 300, 641), K with 4 / 5 / 9 chunks (the
fewest the ring takes, chunk counts that are not a multiple of the stage count, odd counts: the last step runs without a
read), K that is not a whole number of chunks, N that is not a multiple of 128 / 16 / 4 (partial last feature tile, byte stores), with and without the fused ReLU; the
tiled kernel (variant 11) must give the same bytes.  The profile hooks confirm which kernel ran."""
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


SHAPES = [(1000, 9216, 4096), (1000, 4096, 4096), (500, 9216, 4096), (257, 512, 512), (300, 640, 1000), (641, 1152, 520),
          (384, 2048, 4094), (1000, 1024, 1024),
          (600, 2000, 2048)]  # K % 128 != 0: the last chunk of a row runs into the next row's bytes, against zero-padded weights


def _run(gpu, c, relu, variant):
    lib = abi.lib()
    abi.ck(lib.i8ie_ctx_set_option(gpu.h, 2, variant))
    try:
        return gpu.layer_forward_fused("linear", c["q_in"], c["qw"], c["qb"], c["s_in"], c["zp_in"], c["s_w"],
                                       c["s_out"], c["zp_out"], relu=relu)
    finally:
        abi.ck(lib.i8ie_ctx_set_option(gpu.h, 2, 0))


@pytest.mark.parametrize("mkn", SHAPES)
@pytest.mark.parametrize("relu", [False, True])
def test_mlin_bit_exact(gpu, orc, mkn, relu):
    m, k, n = mkn
    c = synth.linear_case(orc, 900 + m + k + n, m, k, n)
    (out, acc, _), names = _kernels_run(gpu, lambda: _run(gpu, c, relu, 0 if n >= 2048 else 83))  # (83: below the automatic thresholds)
    assert any(nm.startswith("mlin_") for nm in names), names
    want = orc.relu(c["out"], c["zp_out"]) if relu else c["out"]
    assert np.array_equal(acc, c["acc"]) and np.array_equal(out, want)
    (out2, acc2, _), names2 = _kernels_run(gpu, lambda: _run(gpu, c, relu, 11))
    assert not any(nm.startswith("mlin") for nm in names2), names2
    assert np.array_equal(out2, out) and np.array_equal(acc2, acc)


@pytest.mark.parametrize("mkn", SHAPES)
def test_mlin_both_row_tiles_bit_exact(gpu, orc, mkn):
    """The 64-row and the 128-row block tile on every shape (variants 84 / 85; automatic: 64 rows when 128-row tiles would give at
    most half the CUs a block, e.g. fc6 / fc7 at 500 rows)."""
    m, k, n = mkn
    c = synth.linear_case(orc, 900 + m + k + n, m, k, n)
    want = orc.relu(c["out"], c["zp_out"])
    for variant, name in ((84, "mlin_64x128"), (85, "mlin_128x128")):
        (out, acc, _), names = _kernels_run(gpu, lambda: _run(gpu, c, True, variant))
        assert name in names, names
        assert np.array_equal(acc, c["acc"]) and np.array_equal(out, want), variant


def test_mlin_picks_the_row_tile_by_block_count(gpu, orc):
    c = synth.linear_case(orc, 5, 500, 1024, 4096)
    (_, _, _), names = _kernels_run(gpu, lambda: _run(gpu, c, False, 0))
    assert "mlin_64x128" in names, names
    c = synth.linear_case(orc, 6, 1000, 1024, 4096)
    (_, _, _), names = _kernels_run(gpu, lambda: _run(gpu, c, False, 0))
    assert "mlin_128x128" in names, names


def test_mlin_extreme_values(gpu, orc):
    """All-255 activations against -128 / +127 weights: the largest accumulators K = 9216 can produce."""
    rng = np.random.default_rng(9)
    n = 640
    q_in = np.full((300, 9216), 255, np.uint8)
    qw = np.full((n, 9216), -128, np.int8)
    qw[1::3] = 127
    qw[2::3] = rng.integers(-128, 128, (len(range(2, n, 3)), 9216))
    qb = rng.integers(-128, 128, n).astype(np.int8)
    for zp_in in (0, 255):
        def run():
            lib = abi.lib()
            abi.ck(lib.i8ie_ctx_set_option(gpu.h, 2, 83))
            try:
                return gpu.layer_forward_fused("linear", q_in, qw, qb, 0.02, zp_in, 0.001, 0.7, 128, relu=False)
            finally:
                abi.ck(lib.i8ie_ctx_set_option(gpu.h, 2, 0))

        (out, acc, _), names = _kernels_run(gpu, run)
        assert any(nm.startswith("mlin_") for nm in names), names
        want, pre, _ = orc.linear(q_in, qw, qb, 0.02, zp_in, 0.001, 0.7, 128, want_acc=True)
        assert np.array_equal(acc, pre) and np.array_equal(out, want)
