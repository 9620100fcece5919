"""ctypes binding of libi8ie_hip.so used by the tests (and a model of the stub an
FFI user would write).  numpy in, numpy out; every call goes
through the C-ABI declared in include/i8ie_hip.h."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# ($I8IE_LIB: another build of the same C-ABI, e.g. the diagnostic build of tools/diag)
LIB_PATH = os.environ.get("I8IE_LIB") or os.path.join(ROOT, "int8inferenceengine_amd", "libi8ie_hip.so")
HEADER = os.path.join(ROOT, "include", "i8ie_hip.h")

_lib = None


def declared_symbols():
    """Every function name declared in include/i8ie_hip.h."""
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(i8ie_[a-z0-9_]+)\s*\(", src)))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libi8ie_hip.so not built: run python -m int8inferenceengine_amd.build")
        _lib = C.CDLL(LIB_PATH)
        _lib.i8ie_last_error.restype = C.c_char_p
        _lib.i8ie_ctx_stream.restype = C.c_void_p
    return _lib


class AbiError(RuntimeError):
    pass


def ck(rc):
    if rc != 0:
        raise AbiError("rc=%d: %s" % (rc, lib().i8ie_last_error().decode()))


class Dev:
    """A device buffer with shape/dtype bookkeeping."""

    def __init__(self, ctx, shape, dtype):
        self.ctx, self.shape, self.dtype = ctx, tuple(shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self.ptr = C.c_void_p()
        ck(lib().i8ie_malloc(ctx.h, C.c_size_t(self.nbytes), C.byref(self.ptr)))

    def get(self):
        out = np.empty(self.shape, self.dtype)
        ck(lib().i8ie_memcpy_d2h(self.ctx.h, out.ctypes.data_as(C.c_void_p), self.ptr, C.c_size_t(self.nbytes)))
        return out

    def free(self):
        if self.ptr:
            lib().i8ie_free(self.ctx.h, self.ptr)
            self.ptr = C.c_void_p()


class Ctx:
    def __init__(self, device=0):
        self.h = C.c_void_p()
        ck(lib().i8ie_ctx_create(device, C.byref(self.h)))

    def close(self):
        if self.h:
            lib().i8ie_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def put(self, arr):
        arr = np.ascontiguousarray(arr)
        d = Dev(self, arr.shape, arr.dtype)
        ck(lib().i8ie_memcpy_h2d(self.h, d.ptr, arr.ctypes.data_as(C.c_void_p), C.c_size_t(d.nbytes)))
        return d

    def empty(self, shape, dtype):
        return Dev(self, shape, dtype)

    def sync(self):
        ck(lib().i8ie_sync(self.h))

    # ---- ops ---------------------------------------------------------------
    def quantize(self, x, scale, zp):
        x = np.ascontiguousarray(x, np.float32)
        d, o = self.put(x), self.empty(x.shape, np.uint8)
        ck(lib().i8ie_quantize_f32_u8(self.h, d.ptr, o.ptr, C.c_int64(x.size), C.c_float(scale), C.c_uint8(zp)))
        r = o.get()
        d.free(); o.free()
        return r

    def dequantize(self, q, scale, zp):
        q = np.ascontiguousarray(q, np.uint8)
        d, o = self.put(q), self.empty(q.shape, np.float32)
        ck(lib().i8ie_dequantize_u8_f32(self.h, d.ptr, o.ptr, C.c_int64(q.size), C.c_float(scale), C.c_uint8(zp)))
        r = o.get()
        d.free(); o.free()
        return r

    def down_scale(self, acc, sa, sb, sc, zp):
        acc = np.ascontiguousarray(acc, np.int32)
        d, o = self.put(acc), self.empty(acc.shape, np.uint8)
        ck(lib().i8ie_down_scale(self.h, d.ptr, o.ptr, C.c_int64(acc.size), C.c_float(sa), C.c_float(sb),
                                 C.c_float(sc), C.c_uint8(zp)))
        r = o.get()
        d.free(); o.free()
        return r

    def relu(self, q, zp):
        q = np.ascontiguousarray(q, np.uint8)
        d, o = self.put(q), self.empty(q.shape, np.uint8)
        ck(lib().i8ie_relu_u8(self.h, d.ptr, o.ptr, C.c_int64(q.size), C.c_uint8(zp)))
        r = o.get()
        d.free(); o.free()
        return r

    def max_pool2d(self, q, k, s):
        q = np.ascontiguousarray(q, np.uint8)
        n, c, h, w = q.shape
        oh, ow = (h - k) // s + 1, (w - k) // s + 1
        d, o = self.put(q), self.empty((n, c, oh, ow), np.uint8)
        ck(lib().i8ie_maxpool2d_u8(self.h, d.ptr, o.ptr, n, c, h, w, k, s))
        r = o.get()
        d.free(); o.free()
        return r

    def conv_offsets(self, qw, qb, s_in, zp_in):
        qw = np.ascontiguousarray(qw, np.int8)
        kc = qw.shape[0]
        K = qw.size // kc
        dw, db, o = self.put(qw), self.put(np.ascontiguousarray(qb, np.int8)), self.empty((kc,), np.int32)
        ck(lib().i8ie_conv_offsets(self.h, dw.ptr, db.ptr, kc, K, C.c_float(s_in), C.c_uint8(zp_in), o.ptr))
        r = o.get()
        dw.free(); db.free(); o.free()
        return r

    def linear_offsets(self, qw, zp_in):
        qw = np.ascontiguousarray(qw, np.int8)
        n, k = qw.shape
        dw, o = self.put(qw), self.empty((n,), np.int32)
        ck(lib().i8ie_linear_offsets(self.h, dw.ptr, n, k, C.c_uint8(zp_in), o.ptr))
        r = o.get()
        dw.free(); o.free()
        return r

    def linear(self, q_in, qw, qb, s_in, zp_in, s_w, s_out, zp_out, want_acc=True):
        """Stateless i8ie_linear_u8s8.  Returns (out, acc or None, oc)."""
        q_in = np.ascontiguousarray(q_in, np.uint8)
        qw = np.ascontiguousarray(qw, np.int8)
        qb = np.ascontiguousarray(qb, np.int8)
        m, k = q_in.shape
        n = qw.shape[0]
        di, dw, db = self.put(q_in), self.put(qw), self.put(qb)
        oc = self.empty((n,), np.int32)
        ck(lib().i8ie_linear_offsets(self.h, dw.ptr, n, k, C.c_uint8(zp_in), oc.ptr))
        out = self.empty((m, n), np.uint8)
        acc = self.empty((m, n), np.int32) if want_acc else None
        ck(lib().i8ie_linear_u8s8(self.h, di.ptr, m, k, dw.ptr, db.ptr, n, oc.ptr, C.c_float(s_in), C.c_float(s_w),
                                  C.c_float(s_out), C.c_uint8(zp_out), out.ptr, acc.ptr if acc else None))
        r = (out.get(), acc.get() if acc else None, oc.get())
        for b in (di, dw, db, oc, out, acc):
            if b is not None:
                b.free()
        return r

    def conv2d(self, q_in, qw, qb, stride, pad, s_in, zp_in, s_w, s_out, zp_out, want_acc=True):
        """Stateless i8ie_conv2d_u8s8.  Returns (out NCHW, acc [n, oh*ow, kc] or None, oc)."""
        q_in = np.ascontiguousarray(q_in, np.uint8)
        qw = np.ascontiguousarray(qw, np.int8)
        qb = np.ascontiguousarray(qb, np.int8)
        n, c, h, w = q_in.shape
        kc, _, kh, kw = qw.shape
        oh, ow = (h - kh + 2 * pad) // stride + 1, (w - kw + 2 * pad) // stride + 1
        di, dw, db = self.put(q_in), self.put(qw), self.put(qb)
        oc = self.empty((kc,), np.int32)
        ck(lib().i8ie_conv_offsets(self.h, dw.ptr, db.ptr, kc, c * kh * kw, C.c_float(s_in), C.c_uint8(zp_in), oc.ptr))
        out = self.empty((n, kc, oh, ow), np.uint8)
        acc = self.empty((n, oh * ow, kc), np.int32) if want_acc else None
        ck(lib().i8ie_conv2d_u8s8(self.h, di.ptr, n, c, h, w, dw.ptr, kc, kh, kw, stride, pad, C.c_uint8(zp_in),
                                  oc.ptr, C.c_float(s_in), C.c_float(s_w), C.c_float(s_out), C.c_uint8(zp_out),
                                  out.ptr, acc.ptr if acc else None))
        r = (out.get(), acc.get() if acc else None, oc.get())
        for b in (di, dw, db, oc, out, acc):
            if b is not None:
                b.free()
        return r

    def layer_forward(self, kind, q_in, qw, qb, s_in, zp_in, s_w, s_out, zp_out, stride=1, pad=0, repeat=1,
                      want_acc=True):
        """Through the layer-handle API (i8ie_*_create / set_output_qparams / forward)."""
        q_in = np.ascontiguousarray(q_in, np.uint8)
        qw = np.ascontiguousarray(qw, np.int8)
        qb = np.ascontiguousarray(qb, np.int8)
        L = C.c_void_p()
        if kind == "linear":
            m, k = q_in.shape
            n = qw.shape[0]
            ck(lib().i8ie_linear_create(self.h, qw.ctypes.data_as(C.c_void_p), qb.ctypes.data_as(C.c_void_p), n, k,
                                        C.c_float(s_w), C.byref(L)))
            oshape, ashape, h, w = (m, n), (m, n), 0, 0
        else:
            m, c, h, w = q_in.shape
            kc, _, kh, kw = qw.shape
            oh, ow = (h - kh + 2 * pad) // stride + 1, (w - kw + 2 * pad) // stride + 1
            ck(lib().i8ie_conv2d_create(self.h, qw.ctypes.data_as(C.c_void_p), qb.ctypes.data_as(C.c_void_p), kc, c,
                                        kh, kw, stride, pad, C.c_float(s_w), C.byref(L)))
            oshape, ashape = (m, kc, oh, ow), (m, oh * ow, kc)
        ck(lib().i8ie_layer_set_output_qparams(L, C.c_float(s_out), C.c_uint8(zp_out)))
        di = self.put(q_in)
        out = self.empty(oshape, np.uint8)
        acc = self.empty(ashape, np.int32) if want_acc else None
        for _ in range(repeat):
            ck(lib().i8ie_layer_forward(L, di.ptr, m, h, w, C.c_float(s_in), C.c_uint8(zp_in), out.ptr,
                                        acc.ptr if acc else None))
        r = (out.get(), acc.get() if acc else None)
        lib().i8ie_layer_destroy(L)
        for b in (di, out, acc):
            if b is not None:
                b.free()
        return r

    # ---- v2 entry points ----------------------------------------------------
    def set_force_fallback(self, on):
        ck(lib().i8ie_ctx_set_option(self.h, 1, 1 if on else 0))

    def layout_convert(self, q, to_nhwc, border=0, fill=0):
        """NCHW numpy -> physical NHWC (with border) numpy, or back."""
        q = np.ascontiguousarray(q, np.uint8)
        if to_nhwc:
            n, c, h, w = q.shape
            oshape = (n, h + 2 * border, w + 2 * border, c)
        else:
            n, hp, wp, c = q.shape
            h, w = hp - 2 * border, wp - 2 * border
            oshape = (n, c, h, w)
        d, o = self.put(q), self.empty(oshape, np.uint8)
        ck(lib().i8ie_layout_convert_u8(self.h, d.ptr, o.ptr, n, c, h, w, 1 if to_nhwc else 0, border, C.c_uint8(fill)))
        r = o.get()
        d.free(); o.free()
        return r

    def max_pool2d_nhwc(self, q_phys, k, s, in_border=0, out_border=0, fill=0):
        q = np.ascontiguousarray(q_phys, np.uint8)
        n, hp, wp, c = q.shape
        h, w = hp - 2 * in_border, wp - 2 * in_border
        oh, ow = (h - k) // s + 1, (w - k) // s + 1
        d, o = self.put(q), self.empty((n, oh + 2 * out_border, ow + 2 * out_border, c), np.uint8)
        ck(lib().i8ie_fill_border_u8(self.h, o.ptr, n, c, oh, ow, out_border, C.c_uint8(fill)))
        ck(lib().i8ie_maxpool2d_u8_nhwc(self.h, d.ptr, in_border, o.ptr, out_border, n, c, h, w, k, s))
        r = o.get()
        d.free(); o.free()
        return r

    @staticmethod
    def to_phys(q_nchw, border, fill):
        """host-side NCHW -> bordered NHWC"""
        n, c, h, w = q_nchw.shape
        p = np.full((n, h + 2 * border, w + 2 * border, c), fill, np.uint8)
        p[:, border:border + h, border:border + w, :] = q_nchw.transpose(0, 2, 3, 1)
        return p

    def layer_forward_fused(self, kind, q_in_nchw, qw, qb, s_in, zp_in, s_w, s_out, zp_out, stride=1, pad=0,
                            in_nhwc=False, out_nhwc=False, relu=False, want_acc=True, in_border=0, out_border=0,
                            flat_chw=None):
        """i8ie_layer_forward_fused.  Input is given in NCHW and laid out on the host as asked (NHWC with a
        zero-point border when in_nhwc); the output is returned in NCHW together with the raw physical array.
        Linear with flat_chw=(c, h, w) and in_nhwc: the [m][c*h*w] rows are handed over as flattened NHWC."""
        q_in = np.ascontiguousarray(q_in_nchw, np.uint8)
        qw = np.ascontiguousarray(qw, np.int8)
        qb = np.ascontiguousarray(qb, np.int8)
        L = C.c_void_p()
        if kind == "linear":
            m, k = q_in.shape
            n = qw.shape[0]
            ck(lib().i8ie_linear_create(self.h, qw.ctypes.data_as(C.c_void_p), qb.ctypes.data_as(C.c_void_p), n, k,
                                        C.c_float(s_w), C.byref(L)))
            oshape, ashape, h, w = (m, n), (m, n), 0, 0
            phys_in = q_in
            if flat_chw is not None and in_nhwc:
                cc, h, w = flat_chw
                phys_in = np.ascontiguousarray(q_in.reshape(m, cc, h, w).transpose(0, 2, 3, 1)).reshape(m, k)
        else:
            m, c, h, w = q_in.shape
            kc, _, kh, kw = qw.shape
            oh, ow = (h - kh + 2 * pad) // stride + 1, (w - kw + 2 * pad) // stride + 1
            ck(lib().i8ie_conv2d_create(self.h, qw.ctypes.data_as(C.c_void_p), qb.ctypes.data_as(C.c_void_p), kc, c,
                                        kh, kw, stride, pad, C.c_float(s_w), C.byref(L)))
            ashape = (m, oh * ow, kc)
            oshape = (m, oh + 2 * out_border, ow + 2 * out_border, kc) if out_nhwc else (m, kc, oh, ow)
            phys_in = self.to_phys(q_in, in_border, zp_in) if in_nhwc else q_in
        ck(lib().i8ie_layer_set_output_qparams(L, C.c_float(s_out), C.c_uint8(zp_out)))
        pref = C.c_int(-1)
        ck(lib().i8ie_layer_preferred_layout(L, C.byref(pref)))
        di = self.put(phys_in)
        out = self.empty(oshape, np.uint8)
        acc = self.empty(ashape, np.int32) if want_acc else None
        if kind != "linear" and out_nhwc and out_border:
            ck(lib().i8ie_fill_border_u8(self.h, out.ptr, m, oshape[3], oshape[1] - 2 * out_border,
                                         oshape[2] - 2 * out_border, out_border, C.c_uint8(zp_out)))
        ck(lib().i8ie_layer_forward_fused(L, di.ptr, 1 if in_nhwc else 0, in_border, m, h, w, C.c_float(s_in),
                                          C.c_uint8(zp_in), 1 if relu else 0, out.ptr, 1 if out_nhwc else 0,
                                          out_border, acc.ptr if acc else None))
        phys = out.get()
        o = phys
        if kind != "linear" and out_nhwc:
            b = out_border
            if b:
                ring = phys.copy()
                ring[:, b:-b, b:-b, :] = zp_out
                assert (ring == zp_out).all(), "output border must hold zp_out"
                o = phys[:, b:-b, b:-b, :]
            o = np.ascontiguousarray(o.transpose(0, 3, 1, 2))
        r = (o, acc.get() if acc else None, pref.value)
        lib().i8ie_layer_destroy(L)
        for bb in (di, out, acc):
            if bb is not None:
                bb.free()
        return r

    def layer_forward_pool(self, q_in_nchw, qw, qb, s_in, zp_in, s_w, s_out, zp_out, stride=1, pad=0, in_nhwc=False,
                           out_nhwc=False, relu=False, in_border=0, out_border=0, pool=None, variant=0, in_s8=False,
                           out_s8=False, names=None):
        """i8ie_layer_forward_pool (pool=(k, s)) or i8ie_layer_forward_fused (pool=None) of a conv layer.
        in_s8 / out_s8: the NHWC input / output in the re-biased form (I8IE_LAYOUT_NHWC_S8: every byte ^ 0x80, borders
        included).  names: list that receives the kernels that ran.  Returns (out NCHW, acc [n, oh*ow, kc])."""
        q_in = np.ascontiguousarray(q_in_nchw, np.uint8)
        qw = np.ascontiguousarray(qw, np.int8)
        qb = np.ascontiguousarray(qb, np.int8)
        m, c, h, w = q_in.shape
        kc, _, kh, kw = qw.shape
        oh, ow = (h - kh + 2 * pad) // stride + 1, (w - kw + 2 * pad) // stride + 1
        ph, pw = (oh, ow) if pool is None else ((oh - pool[0]) // pool[1] + 1, (ow - pool[0]) // pool[1] + 1)
        L = C.c_void_p()
        ck(lib().i8ie_conv2d_create(self.h, qw.ctypes.data_as(C.c_void_p), qb.ctypes.data_as(C.c_void_p), kc, c, kh, kw,
                                    stride, pad, C.c_float(s_w), C.byref(L)))
        ck(lib().i8ie_layer_set_output_qparams(L, C.c_float(s_out), C.c_uint8(zp_out)))
        assert not (in_s8 and not in_nhwc) and not (out_s8 and not out_nhwc)
        phys_in = self.to_phys(q_in, in_border, zp_in) if in_nhwc else q_in
        if in_s8:
            phys_in = phys_in ^ np.uint8(0x80)
        di = self.put(phys_in)
        oshape = (m, ph + 2 * out_border, pw + 2 * out_border, kc) if out_nhwc else (m, kc, ph, pw)
        out = self.empty(oshape, np.uint8)
        acc = self.empty((m, oh * ow, kc), np.int32)
        ozp = zp_out ^ (0x80 if out_s8 else 0)
        if out_nhwc and out_border:
            ck(lib().i8ie_fill_border_u8(self.h, out.ptr, m, kc, ph, pw, out_border, C.c_uint8(ozp)))
        il = (2 if in_s8 else 1) if in_nhwc else 0
        ol = (2 if out_s8 else 1) if out_nhwc else 0

        class _Entry(C.Structure):
            _fields_ = [("name", C.c_char * 64), ("launches", C.c_uint64), ("total_ms", C.c_double),
                        ("total_ops", C.c_double), ("total_bytes", C.c_double)]
        ck(lib().i8ie_ctx_set_option(self.h, 2, variant))
        if names is not None:
            ck(lib().i8ie_profile_start(self.h, 0))
        try:
            if pool is None:
                ck(lib().i8ie_layer_forward_fused(L, di.ptr, il, in_border, m, h, w, C.c_float(s_in),
                                                  C.c_uint8(zp_in), 1 if relu else 0, out.ptr, ol,
                                                  out_border, acc.ptr))
            else:
                ck(lib().i8ie_layer_forward_pool(L, di.ptr, il, in_border, m, h, w, C.c_float(s_in),
                                                 C.c_uint8(zp_in), 1 if relu else 0, pool[0], pool[1], out.ptr,
                                                 ol, out_border, acc.ptr))
        finally:
            ck(lib().i8ie_ctx_set_option(self.h, 2, 0))
            if names is not None:
                ents = (_Entry * 64)()
                cnt = C.c_int(0)
                ck(lib().i8ie_profile_stop(self.h, ents, 64, C.byref(cnt)))
                names.extend(ents[i].name.decode().split("|")[0] for i in range(cnt.value))
        phys = out.get()
        if out_s8:
            phys = phys ^ np.uint8(0x80)
        o = phys
        if out_nhwc:
            b = out_border
            if b:
                ring = phys.copy()
                ring[:, b:-b, b:-b, :] = zp_out
                assert (ring == zp_out).all(), "output border must hold zp_out"
                o = phys[:, b:-b, b:-b, :]
            o = np.ascontiguousarray(o.transpose(0, 3, 1, 2))
        r = (o, acc.get())
        lib().i8ie_layer_destroy(L)
        for bb in (di, out, acc):
            bb.free()
        return r
