"""The deferred-epilogue convolution kernel (tools/diag/csrc/i8ie_dconv.hip, diagnostic build, variant 55) against the oracle,
through the C-ABI: INT32 accumulators and u8 bytes of every output, with and without the folded max-pool, re-biased layouts,
one and two feature passes, short last bands, stride 2, idle blocks, extreme operands and scales that send every dword through
the exact replay.  Run with I8IE_LIB=tools/diag/libi8ie_hip_diag.so (tools/diag/tests/conftest.py)."""
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


DCONV_GEOMS = [
    (300, 96, 27, 27, 256, 5, 1, 2),    # AlexNet conv2: bands of 9 rows, 256-pixel tiles (4 x 32 per wave), 19 K tiles
    (270, 384, 13, 13, 256, 3, 1, 1),   # AlexNet conv5: 169 pixels = 6 tiles of 32 (the last 9 rows full), 27 K tiles
    (257, 384, 13, 13, 256, 3, 1, 1),   # 257 bands: one block runs two
    (70, 256, 20, 20, 256, 3, 1, 1),    # 20 x 20: bands of 12 and 8 rows (short last band: rows past the image are not stored)
    (270, 64, 31, 31, 256, 5, 2, 2),    # stride 2, 16 x 16 outputs, 13 K tiles (one more than the requantiser is spread over)
    (260, 192, 13, 13, 320, 3, 1, 1),   # N = 320: two passes of 256, the second a quarter full (four phases per band)
    (262, 192, 13, 13, 512, 3, 1, 1),   # two full passes
    (200, 224, 13, 13, 256, 3, 1, 1),   # fewer bands than blocks (idle blocks); K = 2016: a K tile with zero-padded chunks
]


@pytest.mark.parametrize("geom", DCONV_GEOMS)
@pytest.mark.parametrize("relu,ob", [(True, 1), (False, 0)])
def test_dconv_bit_exact(gpu, orc, geom, relu, ob):
    """INT32 accumulators and u8 bytes of the kernel that requantises inside its K loop, every output of every image."""
    n, c, h, w, kc, k, stride, pad = geom
    cs = synth.conv_case(orc, 8800 + sum(geom), n, c, h, w, kc, k, stride, pad)
    for in_s8, out_s8 in ((False, False), (True, True)):
        names = []
        out, acc = gpu.layer_forward_pool(cs["q_in"], cs["qw"], cs["qb"], cs["s_in"], cs["zp_in"], cs["s_w"], cs["s_out"],
                                          cs["zp_out"], stride=stride, pad=pad, in_nhwc=True, out_nhwc=True, relu=relu,
                                          in_border=pad, out_border=ob, variant=55, in_s8=in_s8, out_s8=out_s8, names=names)
        assert any(nm.startswith("dconv") for nm in names), names
        assert np.array_equal(acc, cs["acc"]), (in_s8, out_s8)
        assert np.array_equal(out, orc.relu(cs["out"], cs["zp_out"]) if relu else cs["out"]), (in_s8, out_s8)


@pytest.mark.parametrize("geom", [DCONV_GEOMS[0], DCONV_GEOMS[1], DCONV_GEOMS[2], DCONV_GEOMS[3]])
@pytest.mark.parametrize("pool", [(3, 2), (2, 2), (3, 1)])
def test_dconv_folds_the_max_pool(gpu, orc, geom, pool):
    """conv -> relu -> max_pool2d in one launch of the deferred-epilogue kernel: the two feature halves reach the LDS ring
    half a band apart and are pooled in different hand-overs (the second half of a block's last band after its drain)."""
    n, c, h, w, kc, k, stride, pad = geom
    cs = synth.conv_case(orc, 8900 + sum(geom), n, c, h, w, kc, k, stride, pad)
    ref = orc.max_pool2d(orc.relu(cs["out"], cs["zp_out"]), pool[0], pool[1])
    for in_s8, out_s8, ob in ((False, False, 1), (True, True, 2), (False, True, 0)):
        names = []
        out, acc = gpu.layer_forward_pool(cs["q_in"], cs["qw"], cs["qb"], cs["s_in"], cs["zp_in"], cs["s_w"], cs["s_out"],
                                          cs["zp_out"], stride=stride, pad=pad, in_nhwc=True, out_nhwc=True, relu=True,
                                          in_border=pad, out_border=ob, pool=pool, variant=55, in_s8=in_s8, out_s8=out_s8,
                                          names=names)
        assert any(nm.startswith("dconv_pool") for nm in names) and not any(nm.startswith("maxpool") for nm in names), names
        assert np.array_equal(acc, cs["acc"])
        assert np.array_equal(out, ref), (in_s8, out_s8, ob)


def test_dconv_extreme_operands_and_exact_replays(gpu, orc):
    """All-255 activations against +127 / -128 weights (the largest sums), and scales that put many sums exactly ON rounding
    boundaries (0.5 * 0.25 / 0.125: every sum is an integer + zp, the guard sends every dword through the exact sequence)."""
    n, c, h, w, kc, k = 260, 192, 13, 13, 256, 3
    rng = np.random.default_rng(5)
    q_in = np.full((n, c, h, w), 255, np.uint8)
    qw = np.where(rng.random((kc, c, k, k)) < 0.5, 127, -128).astype(np.int8)
    qb = rng.integers(-128, 128, kc).astype(np.int8)
    cases = ((0.02, 3, 0.004, 0.9, 131, q_in, qw),
             (0.5, 128, 0.25, 0.125, 7, rng.integers(126, 131, (n, c, h, w)).astype(np.uint8), (qw // 64).astype(np.int8)))
    for s_in, zp_in, s_w, s_out, zp_out, qi, qwi in cases:
        want, want_acc = orc.conv2d(qi, qwi, qb, 1, 1, np.float32(s_in), zp_in, np.float32(s_w), np.float32(s_out), zp_out, want_acc=True)
        names = []
        out, acc = gpu.layer_forward_pool(qi, qwi, qb, s_in, zp_in, s_w, s_out, zp_out, stride=1, pad=1, in_nhwc=True, out_nhwc=True,
                                          relu=False, in_border=1, out_border=0, variant=55, names=names)
        assert any(nm.startswith("dconv") for nm in names), names
        assert np.array_equal(acc, want_acc.reshape(acc.shape))
        assert np.array_equal(out, want)
