"""ctypes wrapper around oracle/liborc.so (the C restatement in i8ie_oracle.c).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(quiet=True):
    """Compile liborc.so (and oracle/_ref when /root/reference is present)."""
    out = subprocess.DEVNULL if quiet else None
    subprocess.check_call(["make", "-C", _HERE, "all"], stdout=out)
    if os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=out, stderr=out)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liborc.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_quantize_weight.restype = C.c_float
        _LIB.orc_num_threads.restype = C.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def quantize(x, scale, zp):
    x = _c(x, np.float32)
    q = np.empty(x.shape, np.uint8)
    lib().orc_quantize_f32_u8(_p(x), _p(q), C.c_int64(x.size), C.c_float(scale), C.c_uint8(zp))
    return q


def dequantize(q, scale, zp):
    q = _c(q, np.uint8)
    x = np.empty(q.shape, np.float32)
    lib().orc_dequantize_u8_f32(_p(q), _p(x), C.c_int64(q.size), C.c_float(scale), C.c_uint8(zp))
    return x


def down_scale(acc, sa, sb, sc, zp_c):
    acc = _c(acc, np.int32)
    out = np.empty(acc.shape, np.uint8)
    lib().orc_down_scale(_p(out), _p(acc), C.c_int64(acc.size), C.c_float(sa), C.c_float(sb),
                         C.c_float(sc), C.c_uint8(zp_c))
    return out


def relu(q, zp):
    q = _c(q, np.uint8)
    out = np.empty_like(q)
    lib().orc_relu_u8(_p(q), _p(out), C.c_int64(q.size), C.c_uint8(zp))
    return out


def max_pool2d(q, k, s):
    q = _c(q, np.uint8)
    n, c, h, w = q.shape
    oh, ow = (h - k) // s + 1, (w - k) // s + 1
    out = np.empty((n, c, oh, ow), np.uint8)
    lib().orc_max_pool2d_u8(_p(q), _p(out), n, c, h, w, k, s)
    return out


def quantize_weight(w, b):
    w = _c(w, np.float32)
    b = _c(b, np.float32)
    qw = np.empty(w.shape, np.int8)
    qb = np.empty(b.shape, np.int8)
    s = lib().orc_quantize_weight(_p(w), C.c_int64(w.size), _p(b), C.c_int64(b.size), _p(qw), _p(qb))
    return qw, qb, np.float32(s)


def conv_offsets(qw, qb, s_in, zp_in):
    qw = _c(qw, np.int8)
    qb = _c(qb, np.int8)
    kc = qw.shape[0]
    K = qw.size // kc
    oc = np.empty(kc, np.int32)
    lib().orc_conv_offsets(_p(qw), _p(qb), kc, K, C.c_float(s_in), C.c_uint8(zp_in), _p(oc))
    return oc


def linear_offsets(qw, zp_in):
    qw = _c(qw, np.int8)
    n, k = qw.shape
    oc = np.empty(n, np.int32)
    lib().orc_linear_offsets(_p(qw), n, k, C.c_uint8(zp_in), _p(oc))
    return oc


def im2col(img, kh, kw, stride, pad, zp):
    img = _c(img, np.uint8)
    c, h, w = img.shape
    oh, ow = (h - kh + 2 * pad) // stride + 1, (w - kw + 2 * pad) // stride + 1
    M = np.empty((oh * ow, c * kh * kw), np.uint8)
    lib().orc_im2col_u8(_p(M), _p(img), c, h, w, kh, kw, stride, pad, C.c_uint8(zp))
    return M


def gemm_u8s8s32(A, B, oc):
    A = _c(A, np.uint8)
    B = _c(B, np.int8)
    oc = _c(oc, np.int32)
    M, K = A.shape
    N = B.shape[0]
    Cm = np.empty((M, N), np.int32)
    lib().orc_gemm_u8s8s32(M, N, K, _p(A), _p(B), _p(oc), _p(Cm))
    return Cm


def conv2d(q_in, qw, qb, stride, pad, s_in, zp_in, s_w, s_out, zp_out, want_acc=False):
    """Returns (out u8 NCHW, acc int32 [n, oh*ow, kc] or None)."""
    q_in = _c(q_in, np.uint8)
    qw = _c(qw, np.int8)
    qb = _c(qb, np.int8)
    n, c, h, w = q_in.shape
    kc, c2, kh, kw = qw.shape
    assert c2 == c
    oh, ow = (h - kh + 2 * pad) // stride + 1, (w - kw + 2 * pad) // stride + 1
    out = np.empty((n, kc, oh, ow), np.uint8)
    acc = np.empty((n, oh * ow, kc), np.int32) if want_acc else None
    lib().orc_conv2d_u8(_p(q_in), n, c, h, w, _p(qw), _p(qb), kc, kh, kw, stride, pad,
                        C.c_float(s_in), C.c_uint8(zp_in), C.c_float(s_w), C.c_float(s_out),
                        C.c_uint8(zp_out), _p(out), _p(acc) if want_acc else None)
    return out, acc


def linear(q_in, qw, qb, s_in, zp_in, s_w, s_out, zp_out, want_acc=False):
    """Returns (out u8 [m,n], acc_pre_bias, acc_post_bias)."""
    q_in = _c(q_in, np.uint8)
    qw = _c(qw, np.int8)
    qb = _c(qb, np.int8)
    m, k = q_in.shape
    n = qw.shape[0]
    out = np.empty((m, n), np.uint8)
    a0 = np.empty((m, n), np.int32) if want_acc else None
    a1 = np.empty((m, n), np.int32) if want_acc else None
    lib().orc_linear_u8(_p(q_in), m, k, _p(qw), _p(qb), n, C.c_float(s_in), C.c_uint8(zp_in),
                        C.c_float(s_w), C.c_float(s_out), C.c_uint8(zp_out), _p(out),
                        _p(a0) if want_acc else None, _p(a1) if want_acc else None)
    return out, a0, a1


def calib_range(samples, cnt, quantile=1.0):
    s = _c(samples, np.float32).copy()
    assert s.size == 1000
    scale = C.c_float()
    zp = C.c_uint8()
    lib().orc_calib_range(_p(s), C.c_int64(cnt), C.c_float(quantile), C.byref(scale), C.byref(zp))
    return np.float32(scale.value), int(zp.value)


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def num_threads():
    return int(lib().orc_num_threads())


def use_mkl(path=None):
    """Route the contraction of conv2d()/linear() through the reference's own GEMM provider, Intel MKL's
    cblas_gemm_s8u8s32, if its runtime is installed (this image: /opt/conda/lib/libmkl_rt.so.1; $I8IE_MKL_RT
    overrides).  For the timed CPU baseline only.  Returns MKL's version string, or None when the runtime is
    absent or does not reproduce the exact integer results on this host (then nothing changes).  Load it in a
    process that has not imported torch (torch carries its own MKL)."""
    import os

    path = path or os.environ.get("I8IE_MKL_RT", "/opt/conda/lib/libmkl_rt.so.1")
    if not os.path.exists(path):
        return None
    os.environ.setdefault("MKL_THREADING_LAYER", "GNU")  # the oracle's OpenMP is libgomp
    try:
        mkl = C.CDLL(path, mode=C.RTLD_GLOBAL)
        fn = C.cast(mkl.cblas_gemm_s8u8s32, C.c_void_p)
    except (OSError, AttributeError):
        return None
    lib().orc_set_gemm_provider(fn)
    rng = np.random.default_rng(1)
    ok = True
    for (m, k, n, extreme) in ((64, 363, 96, True), (33, 4096, 40, False), (7, 9216, 16, True)):
        a = rng.integers(0, 256, (m, k), dtype=np.uint8)
        b = rng.integers(-128, 128, (n, k), dtype=np.int8)
        if extreme:
            a[...] = 255
            b[::2], b[1::2] = 127, -128
        oc = rng.integers(-1000, 1000, n).astype(np.int32)
        got = np.empty((m, n), np.int32)
        lib().orc_gemm_u8s8s32_provider(m, n, k, _p(a), _p(b), _p(oc), _p(got))
        ok = ok and np.array_equal(got, gemm_u8s8s32(a, b, oc))
    if not ok:  # e.g. an int16-saturating vpmaddubsw path on a non-VNNI / non-Intel host
        lib().orc_set_gemm_provider(None)
        return None
    buf = C.create_string_buffer(256)
    mkl.mkl_get_version_string(buf, 256)
    use_mkl.handle = mkl
    return buf.value.decode().strip()
