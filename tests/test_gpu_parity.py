"""Parity of the HIP path against the oracle and the reference-derived golden
vectors.  Everything here calls through the C-ABI (tests/abi.py, ctypes).
Bar: bit-exact (integer / byte work; fp32 outputs compared as bit patterns)."""
import numpy as np
import pytest

import abi
import synth
from conftest import load_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    c = abi.Ctx(0)
    yield c
    c.close()


# ---------------------------------------------------------------- elementwise --
def test_quantize_golden_and_oracle(gpu, orc):
    for c in load_cases("ref_quantize.npz"):
        assert np.array_equal(gpu.quantize(c["x"], float(c["par"][0]), int(c["par"][1])), c["q"])
    rng = np.random.default_rng(0)
    for n in (1, 15, 16, 17, 4097, 3 * 224 * 224 * 2 + 5):  # ragged tails around the 16-wide vector path
        x = rng.uniform(-4, 4, n).astype(np.float32)
        assert np.array_equal(gpu.quantize(x, 0.025, 127), orc.quantize(x, 0.025, 127))


def test_dequantize_golden_and_oracle(gpu, orc):
    for c in load_cases("ref_dequantize.npz"):
        got = gpu.dequantize(c["q"], float(c["par"][0]), int(c["par"][1]))
        assert np.array_equal(got.view(np.uint32), c["x"].view(np.uint32))
    q = np.random.default_rng(1).integers(0, 256, 100003, dtype=np.uint8)
    assert np.array_equal(gpu.dequantize(q, 0.0371, 9).view(np.uint32), orc.dequantize(q, 0.0371, 9).view(np.uint32))


def test_down_scale_golden(gpu):
    # the fp32 requantiser (src/quantize_utils.cc:27-36): IEEE divide, no contraction
    for c in load_cases("ref_down_scale.npz"):
        sa, sb, sc, zp = c["par"]
        assert np.array_equal(gpu.down_scale(c["acc"], float(sa), float(sb), float(sc), int(zp)), c["out"])


def test_relu_golden_and_oracle(gpu, orc):
    for c in load_cases("ref_relu.npz"):
        assert np.array_equal(gpu.relu(c["q"], int(c["par"][0])), c["out"])
    q = np.random.default_rng(2).integers(0, 256, 96 * 55 * 55 + 7, dtype=np.uint8)
    assert np.array_equal(gpu.relu(q, 131), orc.relu(q, 131))


def test_maxpool_golden_and_oracle(gpu, orc):
    for c in load_cases("ref_maxpool.npz"):
        assert np.array_equal(gpu.max_pool2d(c["q"], int(c["par"][0]), int(c["par"][1])), c["out"])
    q = np.random.default_rng(3).integers(0, 256, (3, 96, 55, 55), dtype=np.uint8)
    assert np.array_equal(gpu.max_pool2d(q, 3, 2), orc.max_pool2d(q, 3, 2))


# ---------------------------------------------------------------- offsets (a4) --
def test_offset_vectors(gpu, orc):
    rng = np.random.default_rng(4)
    qw = rng.integers(-127, 128, (37, 363), dtype=np.int8)
    qb = rng.integers(-127, 128, 37, dtype=np.int8)
    assert np.array_equal(gpu.conv_offsets(qw, qb, np.float32(0.025), 127), orc.conv_offsets(qw, qb, 0.025, 127))
    assert np.array_equal(gpu.linear_offsets(qw, 201), orc.linear_offsets(qw, 201))
    # one-signed weights push the fp32 prefix sums past 2^24: the sequential fp32 order matters
    qw = rng.integers(60, 128, (5, 9216), dtype=np.int8)
    assert np.array_equal(gpu.linear_offsets(qw, 255), orc.linear_offsets(qw, 255))
    assert np.array_equal(gpu.conv_offsets(qw, qb[:5], np.float32(0.01), 255), orc.conv_offsets(qw, qb[:5], 0.01, 255))


# ---------------------------------------------------------------- Linear (a3) --
LINEAR_SHAPES = [(1, 64, 1), (33, 64, 40), (4, 784, 10), (100, 784, 10), (7, 800, 500), (5, 500, 10), (3, 4096, 10),
                 (130, 4096, 300), (64, 9216, 256), (257, 1040, 129)]


@pytest.mark.parametrize("mkn", LINEAR_SHAPES)
def test_linear_stateless_bit_exact(gpu, orc, mkn):
    m, k, n = mkn
    c = synth.linear_case(orc, 100 + m + k + n, m, k, n)
    out, acc, oc = gpu.linear(c["q_in"], c["qw"], c["qb"], c["s_in"], c["zp_in"], c["s_w"], c["s_out"], c["zp_out"])
    assert np.array_equal(oc, orc.linear_offsets(c["qw"], c["zp_in"]))
    assert np.array_equal(acc, c["acc"])  # INT32 pre-requant accumulators
    assert np.array_equal(out, c["out"])


def test_linear_mfma_orientation(gpu, orc):
    # A = "identity" rows with an asymmetric W: a transposed C/D map cannot pass
    k = n = m = 64
    q_in = np.zeros((m, k), np.uint8)
    q_in[np.arange(m), np.arange(k)] = 1
    qw = (np.arange(n)[:, None] * 2 - np.arange(k)[None, :]).clip(-127, 127).astype(np.int8)
    qb = np.zeros(n, np.int8)
    out, acc, _ = gpu.linear(q_in, qw, qb, 1.0, 0, 1.0, 1.0, 100)
    assert np.array_equal(acc, qw.T.astype(np.int32))
    want, _, _ = orc.linear(q_in, qw, qb, 1.0, 0, 1.0, 1.0, 100)
    assert np.array_equal(out, want)


def test_linear_extreme_values(gpu, orc):
    rng = np.random.default_rng(9)
    q_in = np.full((3, 9216), 255, np.uint8)
    qw = np.full((4, 9216), -128, np.int8)
    qw[1] = 127
    qw[2] = rng.integers(-128, 128, 9216)
    qb = np.array([127, -128, 5, 0], np.int8)
    for zp_in in (0, 255):
        out, acc, _ = gpu.linear(q_in, qw, qb, 0.02, zp_in, 0.001, 0.7, 128)
        want, pre, _ = orc.linear(q_in, qw, qb, 0.02, zp_in, 0.001, 0.7, 128, want_acc=True)
        assert np.array_equal(acc, pre) and np.array_equal(out, want)


# ---------------------------------------------------------------- Conv2d (a2) --
CONV_GEOMS = [
    # n, c, h, w, kc, k, stride, pad
    (2, 10, 22, 22, 20, 3, 1, 0), (2, 10, 22, 22, 20, 3, 1, 1), (2, 10, 50, 50, 20, 3, 7, 3),  # unittest/test_layers.py
    (2, 3, 224, 224, 96, 11, 4, 2),   # AlexNet conv1
    (2, 96, 27, 27, 256, 5, 1, 2),    # conv2
    (3, 256, 13, 13, 384, 3, 1, 1),   # conv3
    (2, 384, 13, 13, 384, 3, 1, 1),   # conv4
    (2, 384, 13, 13, 256, 3, 1, 1),   # conv5
    (3, 3, 32, 32, 20, 5, 1, 0), (3, 20, 28, 28, 50, 5, 1, 0), (3, 50, 12, 12, 120, 5, 1, 0),  # simple_conv
    (4, 1, 28, 28, 20, 5, 1, 0), (4, 20, 12, 12, 50, 5, 1, 0),  # two_conv
    (1, 1, 5, 5, 1, 5, 1, 0), (1, 2, 7, 9, 3, 3, 2, 2), (5, 7, 9, 6, 33, 1, 1, 0),  # degenerate / ragged
]


@pytest.mark.parametrize("geom", CONV_GEOMS)
def test_conv2d_stateless_bit_exact(gpu, orc, geom):
    n, c, h, w, kc, k, stride, pad = geom
    cs = synth.conv_case(orc, 7 + sum(geom), n, c, h, w, kc, k, stride, pad)
    out, acc, oc = gpu.conv2d(cs["q_in"], cs["qw"], cs["qb"], stride, pad, cs["s_in"], cs["zp_in"], cs["s_w"],
                              cs["s_out"], cs["zp_out"])
    assert np.array_equal(oc, orc.conv_offsets(cs["qw"], cs["qb"], cs["s_in"], cs["zp_in"]))
    assert np.array_equal(acc, cs["acc"])
    assert np.array_equal(out, cs["out"])


def test_conv2d_padding_uses_input_zero_point(gpu, orc):
    # src/conv2d.cc:24-28: out-of-bounds taps read zero_point, not 0
    for zp_in in (0, 3, 255):
        cs = synth.conv_case(orc, 77 + zp_in, 2, 5, 9, 9, 8, 3, 1, 1, zp_in=zp_in)
        out, acc, _ = gpu.conv2d(cs["q_in"], cs["qw"], cs["qb"], 1, 1, cs["s_in"], zp_in, cs["s_w"], cs["s_out"],
                                 cs["zp_out"])
        assert np.array_equal(acc, cs["acc"]) and np.array_equal(out, cs["out"])


# ---------------------------------------------------------------- layer handles --
def test_layer_handles_cache_offsets(gpu, orc):
    cs = synth.conv_case(orc, 11, 3, 16, 13, 13, 24, 3, 1, 1)
    out, acc = gpu.layer_forward("conv", cs["q_in"], cs["qw"], cs["qb"], cs["s_in"], cs["zp_in"], cs["s_w"],
                                 cs["s_out"], cs["zp_out"], stride=1, pad=1, repeat=3)
    assert np.array_equal(acc, cs["acc"]) and np.array_equal(out, cs["out"])
    ls = synth.linear_case(orc, 12, 50, 800, 500)
    out, acc = gpu.layer_forward("linear", ls["q_in"], ls["qw"], ls["qb"], ls["s_in"], ls["zp_in"], ls["s_w"],
                                 ls["s_out"], ls["zp_out"], repeat=2)
    assert np.array_equal(acc, ls["acc"]) and np.array_equal(out, ls["out"])


def test_layer_handle_rekeys_on_new_input_qparams(gpu, orc):
    import ctypes as C

    cs = synth.conv_case(orc, 13, 2, 8, 10, 10, 12, 3, 1, 1)
    lib = abi.lib()
    L = C.c_void_p()
    abi.ck(lib.i8ie_conv2d_create(gpu.h, cs["qw"].ctypes.data_as(C.c_void_p), cs["qb"].ctypes.data_as(C.c_void_p), 12,
                                  8, 3, 3, 1, 1, C.c_float(cs["s_w"]), C.byref(L)))
    abi.ck(lib.i8ie_layer_set_output_qparams(L, C.c_float(cs["s_out"]), C.c_uint8(cs["zp_out"])))
    di = gpu.put(cs["q_in"])
    out = gpu.empty((2, 12, 10, 10), np.uint8)
    for s_in, zp_in in ((0.025, 127), (0.05, 3), (0.025, 127)):
        abi.ck(lib.i8ie_layer_forward(L, di.ptr, 2, 10, 10, C.c_float(s_in), C.c_uint8(zp_in), out.ptr, None))
        want, _ = orc.conv2d(cs["q_in"], cs["qw"], cs["qb"], 1, 1, np.float32(s_in), zp_in, cs["s_w"], cs["s_out"],
                             cs["zp_out"])
        assert np.array_equal(out.get(), want)
    lib.i8ie_layer_destroy(L)


def test_argument_errors_are_reported(gpu):
    import ctypes as C

    lib = abi.lib()
    d = gpu.empty((16,), np.uint8)
    assert lib.i8ie_relu_u8(gpu.h, None, d.ptr, C.c_int64(16), C.c_uint8(0)) == -1
    assert lib.i8ie_maxpool2d_u8(gpu.h, d.ptr, d.ptr, 1, 1, 4, 4, 5, 1) == -1  # window larger than input
    L = C.c_void_p()
    q = np.zeros(9, np.int8)
    assert lib.i8ie_conv2d_create(gpu.h, q.ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p), 1, 1, 3, 3, 0, 0,
                                  C.c_float(1), C.byref(L)) == -1  # stride 0 (include/conv2d.h:12-14)
    assert b"stride" in lib.i8ie_last_error()


# ---------------------------------------------------------------- batch invariance at full size --
def test_conv_batch_invariance_at_bench_size(gpu, orc):
    """BASELINE size (batch 1000, conv3 geometry): images are independent, so any
    slice of the big batch must equal the same images run alone (and those are
    oracle-checked).  Exercises the chunked im2col path and multi-image tiles."""
    n = 1000
    cs = synth.conv_case(orc, 21, 4, 256, 13, 13, 384, 3, 1, 1)
    rng = np.random.default_rng(5)
    big = rng.integers(0, 256, (n, 256, 13, 13), dtype=np.uint8)
    big[:4] = cs["q_in"]
    big[-4:] = cs["q_in"]
    out, _ = gpu.layer_forward("conv", big, cs["qw"], cs["qb"], cs["s_in"], cs["zp_in"], cs["s_w"], cs["s_out"],
                               cs["zp_out"], stride=1, pad=1, want_acc=False)
    assert np.array_equal(out[:4], cs["out"]) and np.array_equal(out[-4:], cs["out"])
    mid, _ = gpu.layer_forward("conv", big[500:507], cs["qw"], cs["qb"], cs["s_in"], cs["zp_in"], cs["s_w"],
                               cs["s_out"], cs["zp_out"], stride=1, pad=1, want_acc=False)
    assert np.array_equal(out[500:507], mid)


# ================================================================= v2 paths =====
# i8ie_layer_forward_fused: implicit-GEMM conv over NHWC (path A), grouped small-C first
# layer (path B), im2col fallback (path F), fused relu, all layout combinations.
FUSED_GEOMS = [
    (2, 3, 224, 224, 96, 11, 4, 2),   # AlexNet conv1 -> path B
    (2, 96, 27, 27, 256, 5, 1, 2),    # conv2 -> path A
    (3, 256, 13, 13, 384, 3, 1, 1),   # conv3
    (2, 384, 13, 13, 256, 3, 1, 1),   # conv5
    (3, 16, 9, 11, 24, 3, 2, 1),      # path A, ragged N (24 % 16 != 0), stride 2, non-square
    (2, 32, 8, 8, 40, 1, 1, 0),       # 1x1
    (2, 48, 7, 7, 16, 7, 1, 3),       # kernel = image, big padding
    (2, 4, 23, 29, 20, 7, 4, 3),      # path B with c = 4, non-square
    (2, 1, 28, 28, 20, 5, 4, 0),      # path B with c = 1
    (2, 10, 22, 22, 20, 3, 1, 1),     # path F (c % 16 != 0, stride 1)
]


@pytest.mark.parametrize("geom", FUSED_GEOMS)
@pytest.mark.parametrize("layouts", [(False, 0, False, 0), (True, 0, True, 0), (False, 0, True, 2), (True, 0, False, 0),
                                     (True, "pad", True, 1), (True, "pad+1", True, 0)])
def test_conv_fused_all_layouts(gpu, orc, geom, layouts):
    n, c, h, w, kc, k, stride, pad = geom
    in_nhwc, ib, out_nhwc, ob = layouts
    ib = {"pad": pad, "pad+1": pad + 1}.get(ib, ib)
    if kc % 16 != 0:
        ob = 0  # bordered NHWC outputs need features % 16 == 0
    cs = synth.conv_case(orc, 31 + sum(geom), n, c, h, w, kc, k, stride, pad)
    for relu in (False, True):
        out, acc, _ = gpu.layer_forward_fused("conv", cs["q_in"], cs["qw"], cs["qb"], cs["s_in"], cs["zp_in"],
                                              cs["s_w"], cs["s_out"], cs["zp_out"], stride=stride, pad=pad,
                                              in_nhwc=in_nhwc, out_nhwc=out_nhwc, relu=relu, in_border=ib,
                                              out_border=ob)
        want = orc.relu(cs["out"], cs["zp_out"]) if relu else cs["out"]
        assert np.array_equal(acc, cs["acc"])
        assert np.array_equal(out, want)


def test_preferred_layout_and_forced_fallback(gpu, orc):
    cs = synth.conv_case(orc, 41, 2, 96, 27, 27, 256, 5, 1, 2)
    try:
        for force in (False, True):
            gpu.set_force_fallback(force)
            out, acc, pref = gpu.layer_forward_fused("conv", cs["q_in"], cs["qw"], cs["qb"], cs["s_in"], cs["zp_in"],
                                                     cs["s_w"], cs["s_out"], cs["zp_out"], stride=1, pad=2, relu=True,
                                                     out_nhwc=True)
            assert pref == (0 if force else 1)
            assert np.array_equal(acc, cs["acc"]) and np.array_equal(out, orc.relu(cs["out"], cs["zp_out"]))
    finally:
        gpu.set_force_fallback(False)


@pytest.mark.parametrize("mkn", [(33, 64, 40), (100, 784, 10), (5, 500, 10), (130, 4096, 300), (64, 9216, 256)])
def test_linear_fused_relu(gpu, orc, mkn):
    m, k, n = mkn
    c = synth.linear_case(orc, 200 + m + k + n, m, k, n)
    for relu in (False, True):
        out, acc, _ = gpu.layer_forward_fused("linear", c["q_in"], c["qw"], c["qb"], c["s_in"], c["zp_in"], c["s_w"],
                                              c["s_out"], c["zp_out"], relu=relu)
        want = orc.relu(c["out"], c["zp_out"]) if relu else c["out"]
        assert np.array_equal(acc, c["acc"]) and np.array_equal(out, want)


def test_requant_fast_path_boundaries(gpu, orc):
    """Scales that put the real-valued result exactly on / next to integer boundaries, both clamps,
    huge and tiny ratios: the fast requantiser must agree with the exact sequence everywhere."""
    rng = np.random.default_rng(17)
    m, k, n = 96, 128, 128
    q_in = rng.integers(0, 256, (m, k), dtype=np.uint8)
    qw = rng.integers(-3, 4, (n, k)).astype(np.int8)
    qw[:16] = 0
    qw[np.arange(16), np.arange(16)] = 1  # first 16 features: accumulator = one input byte (+offsets)
    qb = rng.integers(-20, 20, n).astype(np.int8)
    combos = [(1.0, 1.0, 1.0, 0), (1.0, 1.0, 1.0, 100), (0.5, 1.0, 0.25, 128), (1.0, 0.5, 4.0, 7),
              (0.025, 0.0031, 0.11, 0), (0.1, 0.1, 0.01, 255), (3.0, 7.0, 0.001, 3), (1e-3, 1e-3, 10.0, 200),
              (0.3333333, 0.1428571, 0.0476190, 64), (1.0, 1.0, 3.0, 1), (2.0, 1.0, 1.0, 254), (1.0, 1.0, 256.0, 0)]
    combos += [(float(a), float(b), float(c), int(z)) for a, b, c, z in
               zip(rng.uniform(0.001, 0.2, 12), rng.uniform(0.0005, 0.01, 12), rng.uniform(0.005, 0.5, 12),
                   rng.integers(0, 256, 12))]
    for s_in, s_w, s_out, zp_out in combos:
        for zp_in in (0, 131):
            want, pre, _ = orc.linear(q_in, qw, qb, np.float32(s_in), zp_in, np.float32(s_w), np.float32(s_out),
                                      zp_out, want_acc=True)
            out, acc, _ = gpu.layer_forward_fused("linear", q_in, qw, qb, np.float32(s_in), zp_in, np.float32(s_w),
                                                  np.float32(s_out), zp_out)
            assert np.array_equal(acc, pre)
            assert np.array_equal(out, want), (s_in, s_w, s_out, zp_out, zp_in)


def test_layout_convert_and_nhwc_pool(gpu, orc):
    rng = np.random.default_rng(23)
    for shape in [(2, 96, 55, 55), (3, 256, 13, 13), (1, 16, 5, 7), (2, 3, 9, 9), (5, 130, 3, 2)]:
        q = rng.integers(0, 256, shape, dtype=np.uint8)
        for b in (0, 1, 2):
            nhwc = gpu.layout_convert(q, True, border=b, fill=77)
            assert np.array_equal(nhwc, gpu.to_phys(q, b, 77))
            assert np.array_equal(gpu.layout_convert(nhwc, False, border=b), q)
    for shape, k, s in [((2, 96, 55, 55), 3, 2), ((2, 256, 27, 27), 3, 2), ((3, 256, 13, 13), 3, 2),
                        ((2, 32, 8, 8), 2, 2), ((1, 16, 4, 4), 2, 1), ((1, 16, 4, 4), 1, 2), ((2, 48, 7, 9), 3, 3)]:
        q = rng.integers(0, 256, shape, dtype=np.uint8)
        want = orc.max_pool2d(q, k, s)
        for ib, ob in ((0, 0), (0, 2), (1, 1), (2, 0)):
            got = gpu.max_pool2d_nhwc(gpu.to_phys(q, ib, 9), k, s, in_border=ib, out_border=ob, fill=200)
            assert np.array_equal(got, gpu.to_phys(want, ob, 200))


def test_conv_batch_invariance_fused_at_bench_size(gpu, orc):
    n = 1000
    cs = synth.conv_case(orc, 21, 4, 256, 13, 13, 384, 3, 1, 1)
    rng = np.random.default_rng(5)
    big = rng.integers(0, 256, (n, 256, 13, 13), dtype=np.uint8)
    big[:4] = cs["q_in"]
    big[-4:] = cs["q_in"]
    out, _, _ = gpu.layer_forward_fused("conv", big, cs["qw"], cs["qb"], cs["s_in"], cs["zp_in"], cs["s_w"],
                                        cs["s_out"], cs["zp_out"], stride=1, pad=1, in_nhwc=True, out_nhwc=True,
                                        want_acc=False, in_border=1, out_border=1)
    assert np.array_equal(out[:4], cs["out"]) and np.array_equal(out[-4:], cs["out"])


# ---- large-tile contraction kernels (256 x 256 / 256 x 192 block tiles, LDS-DMA staging) --------------
# Selected for M >= 65 281 rows and N > 128 features; every output byte and INT32 accumulator of the whole
# batch against the oracle, for the compiled staging variants (0 = default, 5 = one-stage DMA; the two-stage 256-row
# form lives in the diagnostic build, tools/diag): ragged N (masked feature tiles, scalar store path), K tails that end inside a
# 128-byte K tile, stride 2, bordered outputs, fused relu.
LARGE_GEOMS = [
    (400, 128, 13, 13, 256, 3, 1, 1),  # N = 256: one 256-wide feature tile
    (400, 64, 13, 13, 384, 3, 1, 1),   # N = 384: two 192-wide tiles; K = 576 = 4.5 K tiles
    (400, 32, 13, 13, 320, 3, 1, 1),   # N = 320: second 192-wide tile two-thirds full; K = 288
    (100, 96, 27, 27, 200, 5, 1, 2),   # N = 200 (not a multiple of 16: scalar stores), K = 2400, 5 x 5
    (300, 16, 31, 31, 256, 3, 2, 1),   # stride 2, K = 144 (a little over one K tile)
]


@pytest.mark.parametrize("variant", [0, 5])
@pytest.mark.parametrize("geom", LARGE_GEOMS)
def test_conv_large_tiles_bit_exact(gpu, orc, geom, variant):
    n, c, h, w, kc, k, stride, pad = geom
    cs = synth.conv_case(orc, 77 + sum(geom), n, c, h, w, kc, k, stride, pad)
    lib = abi.lib()
    abi.ck(lib.i8ie_ctx_set_option(gpu.h, 2, variant))
    try:
        for relu, ob, want_acc in ((False, 0, True), (True, 1 if kc % 16 == 0 else 0, False)):
            out, acc, _ = gpu.layer_forward_fused("conv", cs["q_in"], cs["qw"], cs["qb"], cs["s_in"], cs["zp_in"],
                                                  cs["s_w"], cs["s_out"], cs["zp_out"], stride=stride, pad=pad,
                                                  in_nhwc=True, out_nhwc=True, relu=relu, in_border=pad,
                                                  out_border=ob, want_acc=want_acc)
            want = orc.relu(cs["out"], cs["zp_out"]) if relu else cs["out"]
            if want_acc:
                assert np.array_equal(acc, cs["acc"])
            assert np.array_equal(out, want)
    finally:
        abi.ck(lib.i8ie_ctx_set_option(gpu.h, 2, 0))


def test_conv_and_linear_chunked_over_the_32bit_offset_range(orc):
    """Activations beyond the 32-bit offset range run as several launches over whole images / rows.  The limit
    is lowered through $I8IE_IGEMM_CHUNK_BYTES in a child process (it is read once per process) so that the
    chunk loop is walked with small tensors: 7 images per launch for the conv, 26 rows for the Linear."""
    import os
    import subprocess
    import sys

    code = r'''
import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import int8inferenceengine_amd, abi, synth, orc
g = abi.Ctx(0)
cs = synth.conv_case(orc, 5, 23, 32, 9, 9, 48, 3, 1, 1)           # 11*11*32 = 3872 B per bordered image
for relu, ob in ((False, 0), (True, 2)):
    out, acc, _ = g.layer_forward_fused("conv", cs["q_in"], cs["qw"], cs["qb"], cs["s_in"], cs["zp_in"], cs["s_w"],
                                        cs["s_out"], cs["zp_out"], stride=1, pad=1, in_nhwc=True, out_nhwc=True,
                                        relu=relu, in_border=1, out_border=ob)
    want = orc.relu(cs["out"], cs["zp_out"]) if relu else cs["out"]
    assert np.array_equal(acc, cs["acc"]) and np.array_equal(out, want)
ls = synth.linear_case(orc, 6, 70, 1024, 200)                      # 1024 B per row: 26 rows per launch
out, acc = g.layer_forward("linear", ls["q_in"], ls["qw"], ls["qb"], ls["s_in"], ls["zp_in"], ls["s_w"], ls["s_out"],
                           ls["zp_out"])
assert np.array_equal(acc, ls["acc"]) and np.array_equal(out, ls["out"])
print("chunked ok")
'''
    env = dict(os.environ, I8IE_IGEMM_CHUNK_BYTES=str(7 * 3872 + 100))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "chunked ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("mchw_n", [(37, 256, 6, 6, 300), (5, 32, 3, 5, 10), (130, 16, 2, 2, 64), (9, 10, 3, 3, 7)])
@pytest.mark.parametrize("fallback", [False, True])
def test_linear_reads_flattened_nhwc_rows(gpu, orc, mchw_n, fallback):
    """x.reshape(n, -1) of an NHWC activation fed to a Linear layer: the layer permutes its weight panel once
    and reads the rows as they lie; accumulators and outputs equal the reference's flattened-NCHW result.
    (c*h*w = 90 is not a multiple of 16 and the forced fallback transposes the input instead.)"""
    m, c, h, w, n = mchw_n
    cs = synth.linear_case(orc, 300 + sum(mchw_n), m, c * h * w, n)
    gpu.set_force_fallback(fallback)
    try:
        for relu in (False, True):
            out, acc, _ = gpu.layer_forward_fused("linear", cs["q_in"], cs["qw"], cs["qb"], cs["s_in"], cs["zp_in"],
                                                  cs["s_w"], cs["s_out"], cs["zp_out"], in_nhwc=True, relu=relu,
                                                  flat_chw=(c, h, w))
            want = orc.relu(cs["out"], cs["zp_out"]) if relu else cs["out"]
            assert np.array_equal(acc, cs["acc"])
            assert np.array_equal(out, want)
    finally:
        gpu.set_force_fallback(False)


@pytest.mark.parametrize("mkn", [(125, 4096, 10), (1, 64, 1), (33, 1024, 16), (7, 784 // 16 * 16, 10), (300, 9216, 3)])
def test_linear_small_n_head_and_fused_dequantize(gpu, orc, mkn):
    """Classifier heads (N <= 16) run one wave per row with v_dot4_i32_i8 + a wavefront reduction; the INT32
    accumulators, the u8 outputs and the fused dequantize(layer(x)) all equal the oracle's sequence."""
    import ctypes as C

    m, k, n = mkn
    cs = synth.linear_case(orc, 900 + m + k + n, m, k, n)
    for relu in (False, True):
        out, acc, _ = gpu.layer_forward_fused("linear", cs["q_in"], cs["qw"], cs["qb"], cs["s_in"], cs["zp_in"],
                                              cs["s_w"], cs["s_out"], cs["zp_out"], relu=relu)
        want = orc.relu(cs["out"], cs["zp_out"]) if relu else cs["out"]
        assert np.array_equal(acc, cs["acc"])
        assert np.array_equal(out, want)
    # dequantize(linear(x)) in one call, with and without the u8 side output
    lib = abi.lib()
    L = C.c_void_p()
    qw, qb = np.ascontiguousarray(cs["qw"]), np.ascontiguousarray(cs["qb"])
    abi.ck(lib.i8ie_linear_create(gpu.h, qw.ctypes.data_as(C.c_void_p), qb.ctypes.data_as(C.c_void_p), n, k,
                                  C.c_float(cs["s_w"]), C.byref(L)))
    abi.ck(lib.i8ie_layer_set_output_qparams(L, C.c_float(cs["s_out"]), C.c_uint8(cs["zp_out"])))
    di = gpu.put(cs["q_in"])
    o8, of = gpu.empty((m, n), np.uint8), gpu.empty((m, n), np.float32)
    want_f = orc.dequantize(cs["out"], cs["s_out"], cs["zp_out"])
    for with_u8 in (False, True):
        abi.ck(lib.i8ie_layer_forward_dequant(L, di.ptr, 0, m, 0, 0, C.c_float(cs["s_in"]), C.c_uint8(cs["zp_in"]), 0,
                                              o8.ptr if with_u8 else None, of.ptr))
        assert np.array_equal(of.get().view(np.uint32), want_f.view(np.uint32))
        if with_u8:
            assert np.array_equal(o8.get(), cs["out"])
    lib.i8ie_layer_destroy(L)
    for d in (di, o8, of):
        d.free()


def test_forward_dequant_general_linear_needs_u8_buffer(gpu, orc):
    import ctypes as C

    cs = synth.linear_case(orc, 77, 9, 64, 40)  # 40 features: ordinary forward + dequantize kernel
    lib = abi.lib()
    L = C.c_void_p()
    abi.ck(lib.i8ie_linear_create(gpu.h, cs["qw"].ctypes.data_as(C.c_void_p), cs["qb"].ctypes.data_as(C.c_void_p), 40,
                                  64, C.c_float(cs["s_w"]), C.byref(L)))
    abi.ck(lib.i8ie_layer_set_output_qparams(L, C.c_float(cs["s_out"]), C.c_uint8(cs["zp_out"])))
    di = gpu.put(cs["q_in"])
    o8, of = gpu.empty((9, 40), np.uint8), gpu.empty((9, 40), np.float32)
    rc = lib.i8ie_layer_forward_dequant(L, di.ptr, 0, 9, 0, 0, C.c_float(cs["s_in"]), C.c_uint8(cs["zp_in"]), 0, None,
                                        of.ptr)
    assert rc == -1 and b"u8 output buffer" in lib.i8ie_last_error()
    abi.ck(lib.i8ie_layer_forward_dequant(L, di.ptr, 0, 9, 0, 0, C.c_float(cs["s_in"]), C.c_uint8(cs["zp_in"]), 0,
                                          o8.ptr, of.ptr))
    assert np.array_equal(o8.get(), cs["out"])
    assert np.array_equal(of.get().view(np.uint32),
                          orc.dequantize(cs["out"], cs["s_out"], cs["zp_out"]).view(np.uint32))
    lib.i8ie_layer_destroy(L)
    for d in (di, o8, of):
        d.free()


def test_gpu_contraction_against_the_reference_gemm_provider(gpu):
    """tests/golden/mkl_gemm_s8u8s32.npz holds results of MKL's cblas_gemm_s8u8s32 -- the routine the reference
    calls -- on seeded operands.  With zp_in = 0 the layer's own offset vector is zero, so the product's INT32
    accumulators for (A, B) must equal MKL's C minus the fixture's oc, bit for bit (MFMA contraction, split-K,
    the classifier-head dot4 kernel and the padded-K fallback all get exercised by the case shapes)."""
    n = 0
    for c in load_cases("mkl_gemm_s8u8s32.npz"):
        A, B, oc, Cm = c["A"], c["B"], c["oc"], c["C"]
        qb = np.zeros(B.shape[0], np.int8)
        _, acc, got_oc = gpu.linear(A, B, qb, np.float32(1.0), 0, np.float32(1.0), np.float32(1.0), 0)
        assert not got_oc.any()
        assert np.array_equal(acc, Cm - oc[None, :])
        out2, acc2, _ = gpu.layer_forward_fused("linear", A, B, qb, np.float32(1.0), 0, np.float32(1.0),
                                                np.float32(1.0), 0)
        assert np.array_equal(acc2, Cm - oc[None, :])
        n += 1
    assert n == 10


def test_ctx_on_a_cu_masked_stream_with_a_cu_limit(orc):
    """I8IE_OPT_CU_LIMIT: a ctx on a stream that may use 64 of the CUs (hipExtStreamCreateWithCUMask) sizes its one-block-per-CU
    kernels for 64 CUs; conv (patch-stationary kernel, pool folded in) and the first-stage kernel stay bit-exact, and a negative
    limit is an argument error."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    mask = (C.c_uint32 * 8)(0xFFFFFFFF, 0xFFFFFFFF, 0, 0, 0, 0, 0, 0)
    stream = C.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(C.byref(stream), C.c_uint32(8), mask) == 0
    lib = abi.lib()
    g = abi.Ctx.__new__(abi.Ctx)
    g.h = C.c_void_p()
    abi.ck(lib.i8ie_ctx_create_on_stream(0, stream, C.byref(g.h)))
    try:
        assert lib.i8ie_ctx_set_option(g.h, 4, -1) != 0
        abi.ck(lib.i8ie_ctx_set_option(g.h, 4, 64))
        cs = synth.conv_case(orc, 77, 300, 96, 27, 27, 256, 5, 1, 2)
        names = []
        out, acc = g.layer_forward_pool(cs["q_in"], cs["qw"], cs["qb"], cs["s_in"], cs["zp_in"], cs["s_w"], cs["s_out"], cs["zp_out"],
                                        stride=1, pad=2, in_nhwc=True, out_nhwc=True, relu=True, in_border=2, pool=(3, 2), names=names)
        assert any(nm.startswith("pconv_pool") for nm in names), names
        assert np.array_equal(acc, cs["acc"])
        assert np.array_equal(out, orc.max_pool2d(orc.relu(cs["out"], cs["zp_out"]), 3, 2))
        cs = synth.conv_case(orc, 78, 150, 3, 67, 83, 64, 7, 4, 3)
        names = []
        out, acc = g.layer_forward_pool(cs["q_in"], cs["qw"], cs["qb"], cs["s_in"], cs["zp_in"], cs["s_w"], cs["s_out"], cs["zp_out"],
                                        stride=4, pad=3, out_nhwc=True, relu=True, out_border=1, pool=(3, 2), names=names)
        assert "stem_conv_pool" in names, names
        assert np.array_equal(acc, cs["acc"])
        assert np.array_equal(out, orc.max_pool2d(orc.relu(cs["out"], cs["zp_out"]), 3, 2))
    finally:
        g.close()
        hip.hipStreamDestroy(stream)
