"""The persistent ping-pong convolution kernel (csrc/i8ie_pp.hip) against the oracle, through the C-ABI.

Every output byte of the whole batch is compared (the kernel has no INT32 dump: accumulators are pinned on
the tiled kernel, which shares the contraction's definition).  Geometries are chosen to hit: one and two
256-wide feature tiles (the second one partly empty), K tails inside a 128-byte K tile, the shortest K the
kernel accepts (2 K tiles), stride 2, tile counts that leave blocks with 0 / 1 / several tiles, M tails,
bordered and plain outputs, with and without the fused ReLU, and both requantiser modes (the proven estimate
and the guarded one).  The profile hooks confirm that the pp kernel is the one that ran."""
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


PP_GEOMS = [
    # n, c, h, w, kc, k, stride, pad
    (400, 128, 13, 13, 256, 3, 1, 1),   # 265 tiles on 256 blocks: most blocks one tile, a few two
    (1300, 32, 13, 13, 256, 3, 1, 1),   # 859 tiles: 3-4 tiles per block, K = 288 (3 K tiles, tail inside the last)
    (400, 64, 13, 13, 384, 3, 1, 1),    # N = 384: second feature tile half empty; K = 576 = 4.5 K tiles
    (400, 32, 13, 13, 320, 3, 1, 1),    # N = 320
    (300, 16, 31, 31, 256, 3, 2, 1),    # stride 2, K = 144: exactly two K tiles (the minimum)
    (100, 96, 27, 27, 256, 5, 1, 2),    # AlexNet conv2 geometry at batch 100: 285 tiles, K = 2400 (19 K tiles)
    (97, 48, 14, 14, 144, 3, 1, 1),     # M = 19012: 75 tiles (181 idle blocks), N = 144 (tile mostly empty), M tail
]


@pytest.mark.parametrize("geom", PP_GEOMS)
@pytest.mark.parametrize("relu,ob", [(True, 1), (False, 0), (True, 0), (False, 2)])
def test_pp_conv_bit_exact(gpu, orc, geom, relu, ob):
    n, c, h, w, kc, k, stride, pad = geom
    cs = synth.conv_case(orc, 1234 + sum(geom), n, c, h, w, kc, k, stride, pad)

    def run():
        lib = abi.lib()
        abi.ck(lib.i8ie_ctx_set_option(gpu.h, 2, 20))  # the pp kernel is an opt-in variant
        try:
            return gpu.layer_forward_fused("conv", cs["q_in"], cs["qw"], cs["qb"], cs["s_in"], cs["zp_in"], cs["s_w"],
                                           cs["s_out"], cs["zp_out"], stride=stride, pad=pad, in_nhwc=True,
                                           out_nhwc=True, relu=relu, in_border=pad, out_border=ob, want_acc=False)
        finally:
            abi.ck(lib.i8ie_ctx_set_option(gpu.h, 2, 0))

    (out, _, _), names = _kernels_run(gpu, run)  # (the harness also checks that border bytes stay zp_out)
    assert any(nm.startswith("pp_conv") for nm in names), names
    want = orc.relu(cs["out"], cs["zp_out"]) if relu else cs["out"]
    assert np.array_equal(out, want)


def test_pp_conv_matches_tiled_kernel_on_extreme_operands(gpu, orc):
    """All-255 activations against +127 / -128 weights (the largest accumulators the layer can produce):
    the pp kernel and the tiled kernel (variant 11) must agree byte for byte with the oracle."""
    n, c, h, w, kc, k = 120, 64, 13, 13, 256, 3
    rng = np.random.default_rng(5)
    q_in = np.full((n, c, h, w), 255, np.uint8)
    qw = np.where(rng.random((kc, c, k, k)) < 0.5, 127, -128).astype(np.int8)
    qb = rng.integers(-128, 128, kc).astype(np.int8)
    s_in, zp_in, s_w, s_out, zp_out = 0.02, 3, 0.004, 0.9, 131
    want = orc.conv2d(q_in, qw, qb, 1, 1, np.float32(s_in), zp_in, np.float32(s_w), np.float32(s_out), zp_out)[0]
    lib = abi.lib()
    outs = {}
    for variant in (20, 11):
        abi.ck(lib.i8ie_ctx_set_option(gpu.h, 2, variant))
        try:
            outs[variant] = gpu.layer_forward_fused("conv", q_in, qw, qb, s_in, zp_in, s_w, s_out, zp_out, stride=1,
                                                    pad=1, in_nhwc=True, out_nhwc=True, in_border=1, want_acc=False)[0]
        finally:
            abi.ck(lib.i8ie_ctx_set_option(gpu.h, 2, 0))
    assert np.array_equal(outs[20], want) and np.array_equal(outs[11], want)
