"""Fused first stage (quantize + small-C strided conv (+ relu) (+ max-pool), FP32 NCHW in, NHWC u8 out) through the
C-ABI entries i8ie_layer_forward_f32_input(_pool) and, from u8 input, i8ie_layer_forward_pool, against the oracle's
quantize -> conv2d (-> relu) (-> max_pool2d): u8 outputs and the convolution's INT32 accumulators.  The profile
hooks confirm which kernel ran (csrc/i8ie_stem.hip where it takes the geometry, csrc/i8ie_first.hip otherwise)."""
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


def run_first(gpu, x, qw, qb, stride, pad, q_scale, q_zp, s_w, s_out, zp_out, relu, ob, pool=None, names=None):
    lib = abi.lib()
    n, c, h, w = x.shape
    kc, _, kh, kw = qw.shape
    oh, ow = (h - kh + 2 * pad) // stride + 1, (w - kw + 2 * pad) // stride + 1
    ph, pw = (oh, ow) if pool is None else ((oh - pool[0]) // pool[1] + 1, (ow - pool[0]) // pool[1] + 1)
    L = C.c_void_p()
    abi.ck(lib.i8ie_conv2d_create(gpu.h, qw.ctypes.data_as(C.c_void_p), qb.ctypes.data_as(C.c_void_p), kc, c, kh, kw,
                                  stride, pad, C.c_float(s_w), C.byref(L)))
    abi.ck(lib.i8ie_layer_set_output_qparams(L, C.c_float(s_out), C.c_uint8(zp_out)))
    yes = C.c_int(-1)
    abi.ck(lib.i8ie_layer_accepts_f32_input(L, h, w, C.byref(yes)))
    if not yes.value:
        lib.i8ie_layer_destroy(L)
        return None
    dx = gpu.put(np.ascontiguousarray(x, np.float32))
    out = gpu.empty((n, ph + 2 * ob, pw + 2 * ob, kc), np.uint8)
    acc = gpu.empty((n, oh * ow, kc), np.int32)
    abi.ck(lib.i8ie_fill_border_u8(gpu.h, out.ptr, n, kc, ph, pw, ob, C.c_uint8(zp_out)))
    abi.ck(lib.i8ie_profile_start(gpu.h, 0))
    try:
        if pool is None:
            abi.ck(lib.i8ie_layer_forward_f32_input(L, dx.ptr, n, h, w, C.c_float(q_scale), C.c_uint8(q_zp),
                                                    1 if relu else 0, out.ptr, ob, acc.ptr))
        else:
            abi.ck(lib.i8ie_layer_forward_f32_input_pool(L, dx.ptr, n, h, w, C.c_float(q_scale), C.c_uint8(q_zp),
                                                         1 if relu else 0, pool[0], pool[1], out.ptr, 1, ob, acc.ptr))
    finally:
        ents = (_Entry * 64)()
        cnt = C.c_int(0)
        abi.ck(lib.i8ie_profile_stop(gpu.h, ents, 64, C.byref(cnt)))
        if names is not None:
            names.extend(ents[i].name.decode().split("|")[0] for i in range(cnt.value))
    phys = out.get()
    acc_h = acc.get()
    lib.i8ie_layer_destroy(L)
    dx.free()
    out.free()
    acc.free()
    if ob:
        ring = phys.copy()
        ring[:, ob:-ob, ob:-ob, :] = zp_out
        assert (ring == zp_out).all()
        phys = phys[:, ob:-ob, ob:-ob, :]
    return np.ascontiguousarray(phys.transpose(0, 3, 1, 2)), acc_h


GEOMS = [
    # n, c, h, w, kc, k, stride, pad
    (2, 3, 224, 224, 96, 11, 4, 2),   # AlexNet conv1
    (5, 3, 224, 224, 96, 11, 4, 2),
    (3, 3, 67, 83, 64, 7, 4, 3),      # non-square; 21-pixel rows: strips of 6 rows; two feature groups, K = 192
    (4, 1, 40, 40, 32, 5, 4, 0),      # one channel, no padding, one feature group
    (2, 2, 50, 31, 96, 3, 8, 1),      # stride 8, two channels, K = 48 (two k-steps)
    (3, 3, 35, 35, 160, 11, 4, 5),    # 5 feature tiles: beyond the first-stage kernel, the older kernel takes it
    (600, 3, 43, 43, 64, 7, 4, 3),    # more images than CUs: blocks walk two or three images (ring and patch sequence
                                      # run through the image boundary)
    (100, 3, 43, 43, 64, 7, 4, 3),    # fewer images than CUs: an image is cut into two parts of whole pooled rows, each a
                                      # unit of its own (the images of n <= 5 above: four parts); variant 12 keeps whole images
]
POOLS = [None, (3, 2), (2, 2)]


@pytest.mark.parametrize("variant", [0, 12])
@pytest.mark.parametrize("geom", GEOMS)
def test_fused_first_layer_bit_exact(gpu, orc, geom, variant):
    n, c, h, w, kc, k, stride, pad = geom
    if variant == 12 and n >= 192:
        pytest.skip("whole images per block anyway")
    abi.ck(abi.lib().i8ie_ctx_set_option(gpu.h, 2, variant))
    rng = np.random.default_rng(sum(geom))
    x = rng.uniform(-2.2, 2.6, (n, c, h, w)).astype(np.float32)
    x.flat[::97] = rng.uniform(-9, 9, x.flat[::97].shape)  # some values outside the no-wrap window (unclamped cast)
    q_scale, q_zp = np.float32(0.025), 127
    q_in = orc.quantize(x, q_scale, q_zp)
    cs = synth.conv_case(orc, 5 + sum(geom), n, c, h, w, kc, k, stride, pad, s_in=q_scale, zp_in=q_zp)
    want, want_acc = orc.conv2d(q_in, cs["qw"], cs["qb"], stride, pad, q_scale, q_zp, cs["s_w"], cs["s_out"], cs["zp_out"],
                                want_acc=True)
    for relu, ob in ((False, 0), (True, 2)):
        for pool in POOLS:
            names = []
            got = run_first(gpu, x, cs["qw"], cs["qb"], stride, pad, q_scale, q_zp, cs["s_w"], cs["s_out"], cs["zp_out"],
                            relu, ob, pool, names)
            assert got is not None, "geometry should be supported"
            got, acc = got
            if kc <= 96:  # the kernel under test is the one that ran, with the pool inside it
                assert ("stem_conv_pool" if pool else "stem_conv") in names and "maxpool_u8_nhwc" not in names, names
            assert np.array_equal(acc, want_acc)  # INT32 pre-requant accumulators of the first-stage kernel itself
            ref = orc.relu(want, cs["zp_out"]) if relu else want
            if pool:
                ref = orc.max_pool2d(ref, pool[0], pool[1])
            assert np.array_equal(got, ref), (relu, ob, pool)
    abi.ck(abi.lib().i8ie_ctx_set_option(gpu.h, 2, 0))


@pytest.mark.parametrize("geom", GEOMS[:5])
def test_first_stage_from_u8_input(gpu, orc, geom):
    """The same kernels fed by an already quantised u8 tensor (NCHW, or NHWC as the engine keeps it between layers)
    through i8ie_layer_forward_pool / i8ie_layer_forward_fused, incl. an NCHW result and non-default input qparams."""
    n, c, h, w, kc, k, stride, pad = geom
    cs = synth.conv_case(orc, 17 + sum(geom), n, c, h, w, kc, k, stride, pad, s_in=0.031, zp_in=99)
    # ((1, 2): a subsampling 1 x 1 window is a pool the first-stage kernel does not fold: conv + the max-pool kernel)
    for in_nhwc, out_nhwc, relu, ob, pool in ((False, True, True, 1, (3, 2)), (True, True, False, 0, None),
                                              (False, False, True, 0, (2, 2)), (False, True, True, 1, (1, 2))):
        out, acc = gpu.layer_forward_pool(cs["q_in"], cs["qw"], cs["qb"], cs["s_in"], cs["zp_in"], cs["s_w"], cs["s_out"],
                                          cs["zp_out"], stride=stride, pad=pad, in_nhwc=in_nhwc, out_nhwc=out_nhwc,
                                          relu=relu, out_border=ob, pool=pool)
        ref = orc.relu(cs["out"], cs["zp_out"]) if relu else cs["out"]
        if pool:
            ref = orc.max_pool2d(ref, pool[0], pool[1])
        assert np.array_equal(acc, cs["acc"])
        assert np.array_equal(out, ref), (in_nhwc, out_nhwc, relu, ob, pool)


def test_unsupported_geometries_say_no(gpu, orc):
    for geom in [(1, 3, 32, 32, 20, 5, 1, 0), (1, 16, 13, 13, 32, 3, 4, 1), (1, 3, 64, 64, 48, 5, 4, 2)]:
        n, c, h, w, kc, k, stride, pad = geom
        cs = synth.conv_case(orc, 1, n, c, h, w, kc, k, stride, pad)
        x = np.zeros((n, c, h, w), np.float32)
        assert run_first(gpu, x, cs["qw"], cs["qb"], stride, pad, 0.025, 127, cs["s_w"], cs["s_out"], cs["zp_out"],
                         False, 0) is None
