"""The patch-stationary convolution kernels (csrc/i8ie_pconv.hip: variants 50 and 54 (N = 384 as two passes of 192 instead of one of 384); csrc/i8ie_tconv.hip: variant 70)
against the oracle, through the C-ABI: u8 outputs AND the INT32 pre-requant accumulators (acc_dbg).

Every output byte of the whole batch is compared.  Geometries hit: one feature pass of 256, two passes of 192
(N = 384), two passes of 256 with the second partly empty (N = 320), one pass of 192; row tiles 11 (one ghost
tile), 13, 16; K tails inside a K tile; stride 2; images split into several bands with a short last band;
tile counts that leave blocks with 0 / 1 / several tiles; bordered and plain outputs; with and without the
fused ReLU; patches that fit LDS twice (ring of whole patches) and C = 384 ones that do not (ring of channel
slices, K slice-major, with and without K padding inside a slice).  The profile hooks confirm that the kernel
under test is the one that ran."""
import ctypes as C

import numpy as np
import pytest

import abi
import synth

pytestmark = pytest.mark.gpu

VARIANTS = {50: "pconv", 54: "pconv", 70: "tconv"}


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


GEOMS = [
    # n, c, h, w, kc, k, stride, pad
    (300, 128, 13, 13, 256, 3, 1, 1),   # 300 one-image tiles on 256 blocks: some blocks two tiles
    (130, 32, 13, 13, 256, 3, 1, 1),    # 130 tiles (idle blocks); K = 288 = 18 chunks: 3 K tiles, tail in the last
    (1100, 64, 13, 13, 384, 3, 1, 1),   # N = 384: two passes of 192; 4-5 tiles per block
    (260, 32, 13, 13, 320, 3, 1, 1),    # N = 320: two passes of 256, the second a quarter full
    (70, 32, 31, 31, 256, 3, 2, 1),     # stride 2, 16 x 16 outputs: one band of 256 pixels
    (100, 96, 27, 27, 256, 5, 1, 2),    # AlexNet conv2 geometry: bands of 9 rows, 300 tiles, K = 2400 (19 K tiles)
    (97, 64, 14, 14, 192, 3, 1, 1),     # 196 pixels = 13 row tiles; one pass of 192
    (70, 32, 20, 20, 256, 3, 1, 1),     # 20 x 20: bands of 12 and 8 rows (short last band)
    (270, 384, 13, 13, 256, 3, 1, 1),   # AlexNet conv5 geometry: 90 KB patch -> two channel slices, K padded per slice
    (140, 384, 13, 13, 384, 3, 1, 1),   # AlexNet conv4 geometry: slices and two feature passes; fewer bands than CUs:
                                        # pconv makes every (band, pass) a unit of its own
    (257, 256, 13, 13, 384, 3, 1, 1),   # AlexNet conv3 geometry; 257 tiles: one block runs two
    (66, 512, 13, 13, 192, 1, 1, 0),    # 1 x 1 kernel, K = 512 = 4 K tiles (the shortest K the team kernel takes)
    (125, 384, 13, 13, 256, 3, 1, 1),   # conv5 of a 125-image shard: pconv splits N = 256 into two passes of 128, 250 units
]


@pytest.mark.parametrize("geom", GEOMS)
@pytest.mark.parametrize("relu,ob", [(True, 1), (False, 0), (False, 2)])
@pytest.mark.parametrize("VARIANT", sorted(VARIANTS))
def test_pconv_bit_exact(gpu, orc, geom, relu, ob, VARIANT):
    n, c, h, w, kc, k, stride, pad = geom
    cs = synth.conv_case(orc, 4321 + sum(geom), n, c, h, w, kc, k, stride, pad)

    def run():
        lib = abi.lib()
        abi.ck(lib.i8ie_ctx_set_option(gpu.h, 2, VARIANT))
        try:
            return gpu.layer_forward_fused("conv", cs["q_in"], cs["qw"], cs["qb"], cs["s_in"], cs["zp_in"], cs["s_w"],
                                           cs["s_out"], cs["zp_out"], stride=stride, pad=pad, in_nhwc=True,
                                           out_nhwc=True, relu=relu, in_border=pad, out_border=ob, want_acc=True)
        finally:
            abi.ck(lib.i8ie_ctx_set_option(gpu.h, 2, 0))

    (out, acc, _), names = _kernels_run(gpu, run)  # (the harness also checks that border bytes stay zp_out)
    if not (VARIANT == 70 and k * k * c < 512):  # (the team kernel leaves K < 4 K tiles to the others)
        assert any(nm.startswith(VARIANTS[VARIANT]) for nm in names), names
    want = orc.relu(cs["out"], cs["zp_out"]) if relu else cs["out"]
    # the INT32 pre-requant accumulators of the kernel under test (the cblas_gemm_s8u8s32 result, src/conv2d.cc:131-133)
    assert np.array_equal(acc, cs["acc"])
    assert np.array_equal(out, want)


@pytest.mark.parametrize("VARIANT", sorted(VARIANTS))
def test_pconv_wider_input_border_than_padding(gpu, orc, VARIANT):
    """in_border 2 around a pad-1 convolution: the window origin is shifted into the border, patch rows wrap
    through border pixels that no valid window touches."""
    n, c, h, w, kc, k = 80, 64, 13, 13, 256, 3
    cs = synth.conv_case(orc, 99, n, c, h, w, kc, k, 1, 1)
    lib = abi.lib()
    abi.ck(lib.i8ie_ctx_set_option(gpu.h, 2, VARIANT))
    try:
        out = gpu.layer_forward_fused("conv", cs["q_in"], cs["qw"], cs["qb"], cs["s_in"], cs["zp_in"], cs["s_w"],
                                      cs["s_out"], cs["zp_out"], stride=1, pad=1, in_nhwc=True, out_nhwc=True,
                                      in_border=2, out_border=0, want_acc=True)[:2]
    finally:
        abi.ck(lib.i8ie_ctx_set_option(gpu.h, 2, 0))
    out, acc = out
    assert np.array_equal(acc, cs["acc"])
    assert np.array_equal(out, cs["out"])


@pytest.mark.parametrize("VARIANT", sorted(VARIANTS))
def test_pconv_extreme_operands(gpu, orc, VARIANT):
    """All-255 activations against +127 / -128 weights (the largest accumulators the layer can produce)."""
    n, c, h, w, kc, k = 120, 64, 13, 13, 256, 3
    rng = np.random.default_rng(5)
    q_in = np.full((n, c, h, w), 255, np.uint8)
    qw = np.where(rng.random((kc, c, k, k)) < 0.5, 127, -128).astype(np.int8)
    qb = rng.integers(-128, 128, kc).astype(np.int8)
    s_in, zp_in, s_w, s_out, zp_out = 0.02, 3, 0.004, 0.9, 131
    want, want_acc = orc.conv2d(q_in, qw, qb, 1, 1, np.float32(s_in), zp_in, np.float32(s_w), np.float32(s_out), zp_out,
                                want_acc=True)
    lib = abi.lib()
    abi.ck(lib.i8ie_ctx_set_option(gpu.h, 2, VARIANT))
    try:
        out, acc = gpu.layer_forward_fused("conv", q_in, qw, qb, s_in, zp_in, s_w, s_out, zp_out, stride=1, pad=1,
                                           in_nhwc=True, out_nhwc=True, in_border=1, want_acc=True)[:2]
    finally:
        abi.ck(lib.i8ie_ctx_set_option(gpu.h, 2, 0))
    assert np.array_equal(acc, want_acc)
    assert np.array_equal(out, want)


# ---- max-pool folded into the epilogue, re-biased (I8IE_LAYOUT_NHWC_S8) input and output ---------------------------------
POOL_GEOMS = [
    (270, 384, 13, 13, 256, 3, 1, 1),   # AlexNet conv5: whole 13 x 13 images per tile, pool 3/2 -> 6 x 6
    (300, 96, 27, 27, 256, 5, 1, 2),    # AlexNet conv2: bands of 9 rows, three per image in one block, windows across bands
    (260, 64, 14, 14, 192, 3, 1, 1),    # one pass of 192 (odd feature tile count per wave)
    (257, 256, 13, 13, 384, 3, 1, 1),   # N = 384: one pass of 384
    (70, 32, 20, 20, 256, 3, 1, 1),     # 20 x 20: bands of 12 and 8 rows
    (300, 32, 31, 31, 256, 3, 2, 1),    # stride 2: 16 x 16 outputs
    (125, 384, 13, 13, 256, 3, 1, 1),   # conv5 of a 125-image shard: (image, pass of 128 features) units, each pools its own features
    (100, 64, 13, 13, 384, 3, 1, 1),    # N = 384 with few images: (image, pass of 192) units, pooled per pass
]


@pytest.mark.parametrize("geom", POOL_GEOMS)
@pytest.mark.parametrize("pool", [(3, 2), (2, 2), (3, 1)])
def test_pconv_folds_the_max_pool(gpu, orc, geom, pool):
    """conv -> relu -> max_pool2d as one launch of the patch-stationary kernel: the pooled bytes are those of
    orc.max_pool2d(orc.relu(conv)), the accumulators those of the convolution (unpooled), for windows inside a band
    and across bands; with a bordered result and with re-biased bytes at either end."""
    n, c, h, w, kc, k, stride, pad = geom
    cs = synth.conv_case(orc, 777 + sum(geom), n, c, h, w, kc, k, stride, pad)
    ref = orc.max_pool2d(orc.relu(cs["out"], cs["zp_out"]), pool[0], pool[1])
    for in_s8, out_s8, ob in ((False, False, 1), (True, True, 2), (False, True, 0)):
        names = []
        out, acc = gpu.layer_forward_pool(cs["q_in"], cs["qw"], cs["qb"], cs["s_in"], cs["zp_in"], cs["s_w"], cs["s_out"],
                                          cs["zp_out"], stride=stride, pad=pad, in_nhwc=True, out_nhwc=True, relu=True,
                                          in_border=pad, out_border=ob, pool=pool, variant=50, in_s8=in_s8, out_s8=out_s8,
                                          names=names)
        assert any(nm.startswith("pconv_pool") for nm in names) and not any(nm.startswith("maxpool") for nm in names), names
        assert np.array_equal(acc, cs["acc"])
        assert np.array_equal(out, ref), (in_s8, out_s8, ob)


@pytest.mark.parametrize("geom", POOL_GEOMS[:3])
def test_pconv_rebiased_layouts_without_pool(gpu, orc, geom):
    n, c, h, w, kc, k, stride, pad = geom
    cs = synth.conv_case(orc, 778 + sum(geom), n, c, h, w, kc, k, stride, pad)
    for in_s8, out_s8, relu, ob in ((True, False, False, 0), (False, True, True, 1), (True, True, True, 2)):
        names = []
        out, acc = gpu.layer_forward_pool(cs["q_in"], cs["qw"], cs["qb"], cs["s_in"], cs["zp_in"], cs["s_w"], cs["s_out"],
                                          cs["zp_out"], stride=stride, pad=pad, in_nhwc=True, out_nhwc=True, relu=relu,
                                          in_border=pad, out_border=ob, variant=50, in_s8=in_s8, out_s8=out_s8, names=names)
        assert any(nm.startswith("pconv") for nm in names) and "rebias_u8" not in names, names
        assert np.array_equal(acc, cs["acc"])
        assert np.array_equal(out, orc.relu(cs["out"], cs["zp_out"]) if relu else cs["out"])


def test_pool_and_rebiased_layouts_where_no_kernel_folds_them(gpu, orc):
    """The same calls on launches the patch-stationary kernel declines (a handful of images: the tiled kernel runs) and on
    the any-geometry path: the library pools / re-biases around the kernel, same bytes."""
    n, c, h, w, kc, k, stride, pad = 3, 32, 13, 13, 64, 3, 1, 1
    cs = synth.conv_case(orc, 5150, n, c, h, w, kc, k, stride, pad)
    ref = orc.max_pool2d(orc.relu(cs["out"], cs["zp_out"]), 3, 2)
    for in_s8, out_s8, ob in ((False, False, 0), (True, True, 1), (True, False, 1), (False, True, 0)):
        names = []
        out, acc = gpu.layer_forward_pool(cs["q_in"], cs["qw"], cs["qb"], cs["s_in"], cs["zp_in"], cs["s_w"], cs["s_out"],
                                          cs["zp_out"], stride=stride, pad=pad, in_nhwc=True, out_nhwc=True, relu=True,
                                          in_border=pad, out_border=ob, pool=(3, 2), in_s8=in_s8, out_s8=out_s8, names=names)
        assert not any(nm.startswith("pconv") for nm in names), names
        assert np.array_equal(acc, cs["acc"]) and np.array_equal(out, ref), (in_s8, out_s8, ob)


@pytest.mark.parametrize("geom", [POOL_GEOMS[0], POOL_GEOMS[1], (3, 32, 13, 13, 64, 3, 1, 1)])
def test_subsampling_1x1_pool_is_a_pool_everywhere(gpu, orc, geom):
    """max_pool2d(kernel_size=1, stride=2) subsamples (src/functional.cc:36-64 has no special case for it).  No kernel folds
    it, so on launches the patch-stationary kernel would otherwise take (and pool) the library must run conv + the max-pool
    kernel: the result has the POOLED shape and the oracle's bytes -- round 3 wrote the unpooled tensor into the pooled buffer."""
    n, c, h, w, kc, k, stride, pad = geom
    cs = synth.conv_case(orc, 4242 + sum(geom), n, c, h, w, kc, k, stride, pad)
    ref = orc.max_pool2d(orc.relu(cs["out"], cs["zp_out"]), 1, 2)
    for in_s8, out_s8, ob in ((False, False, 0), (True, True, 1)):
        names = []
        out, acc = gpu.layer_forward_pool(cs["q_in"], cs["qw"], cs["qb"], cs["s_in"], cs["zp_in"], cs["s_w"], cs["s_out"],
                                          cs["zp_out"], stride=stride, pad=pad, in_nhwc=True, out_nhwc=True, relu=True,
                                          in_border=pad, out_border=ob, pool=(1, 2), in_s8=in_s8, out_s8=out_s8, names=names)
        assert not any(nm.startswith("pconv_pool") for nm in names), names
        assert out.shape == ref.shape
        assert np.array_equal(acc, cs["acc"]) and np.array_equal(out, ref), (in_s8, out_s8, ob)
