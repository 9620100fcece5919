"""The oracle (oracle/i8ie_oracle.c) against golden vectors produced by the
reference's own compiled quantize_utils.cc / functional.cc (tests/golden/
make_golden.py).  Bit-exact: integer / byte work, and fp32 outputs compared
as bit patterns."""
import numpy as np
import pytest

from conftest import load_cases


def test_quantize_matches_reference(orc):
    for c in load_cases("ref_quantize.npz"):
        scale, zp = float(c["par"][0]), int(c["par"][1])
        got = orc.quantize(c["x"], scale, zp)
        assert np.array_equal(got, c["q"])


def test_quantize_wraparound_known_answer(orc):
    # unclamped cast wraps (10.0 -> 527 -> 15, -10.0 -> -273 -> 239)
    x = np.array([10, -10, 3.2, -3.3, 0, 0.0124, -0.0126, 5.12], np.float32)
    assert orc.quantize(x, 0.025, 127).tolist() == [15, 239, 255, 251, 127, 127, 126, 75]


def test_dequantize_matches_reference(orc):
    for c in load_cases("ref_dequantize.npz"):
        scale, zp = float(c["par"][0]), int(c["par"][1])
        got = orc.dequantize(c["q"], scale, zp)
        assert np.array_equal(got.view(np.uint32), c["x"].view(np.uint32))


def test_down_scale_matches_reference(orc):
    for c in load_cases("ref_down_scale.npz"):
        sa, sb, sc, zp = c["par"]
        got = orc.down_scale(c["acc"], float(sa), float(sb), float(sc), int(zp))
        assert np.array_equal(got, c["out"])


def test_relu_matches_reference(orc):
    for c in load_cases("ref_relu.npz"):
        assert np.array_equal(orc.relu(c["q"], int(c["par"][0])), c["out"])


def test_maxpool_matches_reference(orc):
    for c in load_cases("ref_maxpool.npz"):
        k, s = int(c["par"][0]), int(c["par"][1])
        assert np.array_equal(orc.max_pool2d(c["q"], k, s), c["out"])


# ---- rows whose reference translation units need mkl.h (parity unpinned by a
# reference run): cross-check the restatement against an independent int64
# formulation of the same definition. -----------------------------------------

def _rand_layer(rng, kc, c, kh, kw):
    w = rng.uniform(-1, 1, (kc, c, kh, kw)).astype(np.float32) * np.float32(np.sqrt(6.0 / (c * kh * kw)))
    b = rng.uniform(-1, 1, kc).astype(np.float32) * np.float32(1.0 / np.sqrt(c * kh * kw))
    return w, b


def test_quantize_weight_definition(orc):
    rng = np.random.default_rng(1)
    w, b = _rand_layer(rng, 20, 10, 3, 3)
    qw, qb, s = orc.quantize_weight(w, b)
    mx = max(w.max(), b.max())
    mn = min(w.min(), b.min())
    s_ref = np.float32(np.float32(mx - mn) / np.float32(127))
    assert s == s_ref
    assert np.array_equal(qw, np.trunc(w / s_ref).astype(np.int32).astype(np.int8))
    assert np.array_equal(qb, np.trunc(b / s_ref).astype(np.int32).astype(np.int8))
    assert np.abs(qw.astype(int)).max() <= 127


@pytest.mark.parametrize("geom", [
    # (n, c, h, w, kc, k, stride, pad)  -- unittest/test_layers.py geometries + AlexNet-like
    (3, 10, 22, 22, 20, 3, 1, 0), (3, 10, 22, 22, 20, 3, 1, 1), (2, 10, 50, 50, 20, 3, 7, 3),
    (2, 3, 47, 47, 16, 11, 4, 2), (2, 8, 27, 27, 24, 5, 1, 2), (2, 16, 13, 13, 12, 3, 1, 1),
    (1, 1, 28, 28, 20, 5, 1, 0),
])
def test_conv2d_against_int64_definition(orc, geom):
    import torch
    import torch.nn.functional as F

    n, c, h, w, kc, k, stride, pad = geom
    rng = np.random.default_rng(sum(geom))
    wf, bf = _rand_layer(rng, kc, c, k, k)
    qw, qb, s_w = orc.quantize_weight(wf, bf)
    s_in, zp_in = np.float32(0.025), 127
    s_out, zp_out = np.float32(0.04), 119
    q_in = rng.integers(0, 256, (n, c, h, w), dtype=np.uint8)
    out, acc = orc.conv2d(q_in, qw, qb, stride, pad, s_in, zp_in, s_w, s_out, zp_out, want_acc=True)
    # independent: pad with zp, exact conv in float64 (all values < 2^53)
    xp = np.full((n, c, h + 2 * pad, w + 2 * pad), zp_in, np.float64)
    xp[:, :, pad:pad + h, pad:pad + w] = q_in
    conv = F.conv2d(torch.from_numpy(xp), torch.from_numpy(qw.astype(np.float64)), stride=stride).numpy()
    oc = orc.conv_offsets(qw, qb, s_in, zp_in)
    wsum = qw.reshape(kc, -1).astype(np.int64).sum(1)
    oc_ref = np.trunc(np.float32(qb.astype(np.float32) / s_in) - (zp_in * wsum).astype(np.float32)).astype(np.int64)
    assert np.array_equal(oc, oc_ref)  # exact here: |zp*sum| < 2^24
    acc_ref = conv.astype(np.int64) + oc.astype(np.int64)[None, :, None, None]  # NCHW
    oh, ow = out.shape[2:]
    assert np.array_equal(acc.reshape(n, oh, ow, kc).transpose(0, 3, 1, 2), acc_ref)
    assert np.array_equal(out, orc.down_scale(acc_ref.astype(np.int32), s_in, s_w, s_out, zp_out))


@pytest.mark.parametrize("mkn", [(4, 784, 10), (7, 800, 500), (5, 500, 10), (3, 4096, 10), (9, 9216, 64)])
def test_linear_against_int64_definition(orc, mkn):
    m, k, n = mkn
    rng = np.random.default_rng(m * k + n)
    wf = rng.uniform(-1, 1, (n, k)).astype(np.float32) * np.float32(np.sqrt(6.0 / k))
    bf = rng.uniform(-1, 1, n).astype(np.float32) * np.float32(1 / np.sqrt(k))
    qw, qb, s_w = orc.quantize_weight(wf, bf)
    s_in, zp_in, s_out, zp_out = np.float32(0.031), 64, np.float32(0.09), 101
    q_in = rng.integers(0, 256, (m, k), dtype=np.uint8)
    out, pre, post = orc.linear(q_in, qw, qb, s_in, zp_in, s_w, s_out, zp_out, want_acc=True)
    oc = orc.linear_offsets(qw, zp_in)
    pre_ref = q_in.astype(np.int64) @ qw.astype(np.int64).T + oc.astype(np.int64)[None, :]
    assert np.array_equal(pre, pre_ref)
    bias_f = (qb.astype(np.float32) / s_in).astype(np.float32)
    post_ref = np.trunc(pre_ref.astype(np.float32) + bias_f[None, :]).astype(np.int64)
    assert np.array_equal(post, post_ref)
    assert np.array_equal(out, orc.down_scale(post_ref.astype(np.int32), s_in, s_w, s_out, zp_out))


def test_im2col_and_gemm_pieces(orc):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (3, 9, 8), dtype=np.uint8)
    M = orc.im2col(img, 3, 3, 2, 1, 200)
    oh, ow = (9 - 3 + 2) // 2 + 1, (8 - 3 + 2) // 2 + 1
    assert M.shape == (oh * ow, 27)
    xp = np.full((3, 11, 10), 200, np.uint8)
    xp[:, 1:10, 1:9] = img
    for r in (0, 5, oh * ow - 1):
        ti, tj = divmod(r, ow)
        assert np.array_equal(M[r], xp[:, ti * 2:ti * 2 + 3, tj * 2:tj * 2 + 3].reshape(-1))
    A = rng.integers(0, 256, (11, 363), dtype=np.uint8)
    A[0, :] = 255
    B = rng.integers(-128, 128, (7, 363), dtype=np.int8)
    B[0, :] = -128
    oc = rng.integers(-1000, 1000, 7).astype(np.int32)
    assert np.array_equal(orc.gemm_u8s8s32(A, B, oc), A.astype(np.int64) @ B.astype(np.int64).T + oc)


def test_calib_range_restatement(orc):
    rng = np.random.default_rng(2)
    s = rng.normal(0.3, 2.0, 1000).astype(np.float32)
    scale, zp = orc.calib_range(s, 1000)
    mn, mx = min(s.min(), 0), max(s.max(), 0)
    zp_ref = int(255 * (0 - float(mn)) / (float(mx) - float(mn) + 1e-9))
    assert zp == zp_ref
    assert scale == np.float32((0 - mn) / np.float32(zp)) if zp else np.float32((mx - mn) / 255)


def test_calib_range_golden(orc):
    """Calibrator::sample + get_range of the reference's own compiled src/calibrator.cc (tests/golden/
    make_golden.py): 18 cases -- quantiles, values fed in several sample() calls, one-sided ranges (zero point
    0 / 255), the all-zero fallback scale 1, an outlier with and without a quantile that drops it."""
    n = 0
    for case in load_cases("ref_calib_range.npz"):
        q = float(case["par"][0])
        scale, zp = orc.calib_range(case["x"].copy(), 1000, q)
        assert np.float32(scale).view(np.uint32) == case["scale"].view(np.uint32) and int(zp) == int(case["zp"])
        n += 1
    assert n == 18


def test_product_calibrator_golden():
    """The same vectors through the calibrator inside _CXX_i8ie (host code of the layers' prepare/convert)."""
    import int8inferenceengine_amd  # noqa: F401
    import _CXX_i8ie as cx

    for case in load_cases("ref_calib_range.npz"):
        q, cuts = float(case["par"][0]), [int(c) for c in case["par"][1:]]
        chunks = np.split(case["x"], cuts) if cuts else [case["x"]]
        scale, zp = cx.calibrator_range([np.ascontiguousarray(c) for c in chunks], q)
        assert np.float32(scale).view(np.uint32) == case["scale"].view(np.uint32) and int(zp) == int(case["zp"])


def test_contraction_against_the_reference_gemm_provider(orc):
    """The reference's contraction is Intel MKL's cblas_gemm_s8u8s32 (src/conv2d.cc:131-133,
    src/fully_connected.cc:39-41).  tests/golden/make_golden_mkl.py called that entry point with the
    reference's argument pattern (row-major, A u8 not transposed, B s8 transposed, CblasRowOffset: oc[j] added
    to column j, alpha 1, beta 0, zero operand offsets) on seeded operands incl. the extreme 255 x (127 | -128)
    case; the oracle's GEMM must reproduce MKL's INT32 results exactly."""
    n = 0
    for c in load_cases("mkl_gemm_s8u8s32.npz"):
        got = orc.gemm_u8s8s32(c["A"], c["B"], c["oc"])
        assert np.array_equal(got, c["C"])
        n += 1
    assert n == 10


def test_contraction_against_live_mkl_when_present(orc):
    """Same check against MKL itself where its runtime is installed (this image: /opt/conda/lib).  Skipped when
    absent, and on non-Intel hosts, where MKL may take a vpmaddubsw path that saturates in int16."""
    import ctypes as C
    import os

    path = os.environ.get("I8IE_MKL_RT", "/opt/conda/lib/libmkl_rt.so.1")
    if not os.path.exists(path) or "GenuineIntel" not in open("/proc/cpuinfo").read():
        pytest.skip("no MKL runtime / not an Intel host")
    os.environ.setdefault("MKL_THREADING_LAYER", "GNU")
    mkl = C.CDLL(path, mode=C.RTLD_GLOBAL)
    rng = np.random.default_rng(9)
    for M, K, N in ((17, 363, 20), (40, 4096, 10), (3, 800, 500)):
        A = rng.integers(0, 256, (M, K), dtype=np.uint8)
        B = rng.integers(-128, 128, (N, K), dtype=np.int8)
        oc = rng.integers(-50000, 50000, N).astype(np.int32)
        Cm = np.empty((M, N), np.int32)
        mkl.cblas_gemm_s8u8s32(101, 111, 112, 171, C.c_int(M), C.c_int(N), C.c_int(K), C.c_float(1.0),
                               A.ctypes.data_as(C.c_void_p), C.c_int(K), C.c_int8(0), B.ctypes.data_as(C.c_void_p),
                               C.c_int(K), C.c_int8(0), C.c_float(0.0), Cm.ctypes.data_as(C.c_void_p), C.c_int(N),
                               oc.ctypes.data_as(C.c_void_p))
        assert np.array_equal(orc.gemm_u8s8s32(A, B, oc), Cm)
