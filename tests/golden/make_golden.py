#!/usr/bin/env python3
"""Generate golden vectors from the reference's OWN compiled code.

Runs only in the build container (needs /root/reference): `make -C oracle ref`
compiles the reference's src/quantize_utils.cc and src/functional.cc where they
lie into oracle/_ref/_i8ie_ref_partial*.so; this script feeds seeded inputs
through it and stores inputs + expected outputs as small .npz fixtures (data
only, no pickles) next to this file.  The fixtures are committed; the reference
itself never travels.

Covers SURVEY.md section 8 rows a1 (quantize), a5 (down_scale), a6 (dequantize),
a7 (relu u8), a8 (max_pool2d u8) and the calibrator's range computation
(src/calibrator.cc, compiled the same way).  Rows a2/a3/a4/a10 live in translation
units that include mkl.h (absent from the image) and cannot be built here.

usage:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
import _i8ie_ref_partial as ref  # noqa: E402

rng = np.random.default_rng(20261004)


ONLY = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--only=")]  # e.g. --only=ref_calib_range.npz


def save(name, cases):
    if ONLY and name not in ONLY:
        return  # keep the committed fixture byte for byte
    flat = {}
    for i, case in enumerate(cases):
        for k, v in case.items():
            flat["%d_%s" % (i, k)] = np.asarray(v)
    flat["n_cases"] = np.asarray(len(cases))
    np.savez_compressed(os.path.join(HERE, name), **flat)
    print(name, len(cases), "cases")


# ---- a1 quantize (src/quantize_utils.cc:44-52) ------------------------------
q_cases = []


def add_q(x, scale, zp):
    x = np.ascontiguousarray(x, np.float32)
    out = ref.quantize(ref.f32(x), scale, zp)
    assert abs(out.scale() - np.float32(scale)) == 0 and out.zero_point() == zp
    q_cases.append(dict(x=x, par=np.array([scale, zp], np.float64), q=out.numpy().copy()))


add_q([10, -10, 3.2, -3.3, 0, 0.0124, -0.0126, 5.12], 0.025, 127)  # wrap-around (no clamp)
add_q(rng.uniform(-1, 1, (4, 4)), 0.025, 100)                        # unittest/test_quantization.py
add_q(rng.uniform(-2.0, 2.4, (2, 3, 32, 32)), 0.025, 127)            # Module.__call__ qparams
add_q(rng.uniform(-8, 8, (3, 1000)), 0.025, 127)                     # partly out of range -> wraps
add_q(rng.uniform(0, 1, (5, 1, 28, 28)), 0.025, 127)                 # MNIST-like
add_q((np.arange(-300, 300, dtype=np.float32) * 0.025), 0.025, 127)  # exact grid points
add_q(rng.normal(0, 1, 4097) * 3, 0.0371, 3)
add_q(rng.normal(0, 50, 1024), 1.0, 0)
save("ref_quantize.npz", q_cases)

# ---- a6 dequantize (src/quantize_utils.cc:38-42,54-58) ----------------------
d_cases = []
for scale, zp, shape in [(0.025, 127, (7, 10)), (0.0371, 3, (1000,)), (1.0, 0, (256,)),
                         (0.1234567, 255, (3, 5, 7)), (3.5e-3, 64, (100, 10))]:
    q = rng.integers(0, 256, shape, dtype=np.uint8)
    out = ref.dequantize(ref.u8(q, scale, zp))
    d_cases.append(dict(q=q, par=np.array([scale, zp], np.float64), x=out.numpy().copy()))
q = np.arange(256, dtype=np.uint8)
d_cases.append(dict(q=q, par=np.array([0.025, 127], np.float64),
                    x=ref.dequantize(ref.u8(q, 0.025, 127)).numpy().copy()))
save("ref_dequantize.npz", d_cases)

# ---- a5 down_scale (src/quantize_utils.cc:27-36) ----------------------------
ds_cases = []


def add_ds(acc, sa, sb, sc, zp):
    acc = np.ascontiguousarray(acc, np.int32)
    out = ref.down_scale(acc, sa, sb, sc, zp)
    ds_cases.append(dict(acc=acc, par=np.array([sa, sb, sc, zp], np.float64), out=out.copy()))


def f32(x):
    return float(np.float32(x))


for sa, sb, sc, zp, span in [
    (0.025, 0.0031, 0.11, 0, 3000), (0.025, 0.00047, 0.05, 121, 400000),
    (0.0438, 0.00091, 0.21, 87, 100000), (0.013, 0.0007, 0.0093, 143, 9000),
    (1.0, 1.0, 1.0, 0, 400), (0.5, 0.25, 0.125, 128, 300), (0.0071, 0.0123, 0.3333, 255, 2000000),
]:
    sa, sb, sc = f32(sa), f32(sb), f32(sc)
    add_ds(rng.integers(-span, span, 20000), sa, sb, sc, zp)
# dense sweep across every output level incl. both clamps (all integers in a window)
sa, sb, sc = f32(0.025), f32(0.002), f32(0.07)
add_ds(np.arange(-250000, 250000, 7), sa, sb, sc, 90)
# |C| > 2^24: int->float conversion itself rounds
add_ds(rng.integers(-(2 ** 30), 2 ** 30, 20000), f32(1e-3), f32(1e-3), f32(2.5), 100)
add_ds(np.array([2 ** 24 + 1, 2 ** 24 + 3, -(2 ** 24) - 1, 2 ** 31 - 1, -(2 ** 31), 0, 1, -1]),
       f32(3e-4), f32(7e-3), f32(0.4), 17)
save("ref_down_scale.npz", ds_cases)

# ---- a7 relu<u8> (src/functional.cc:15-26) ----------------------------------
r_cases = []
for zp, shape in [(0, (33,)), (127, (2, 3, 9, 9)), (255, (100,)), (90, (4, 1000)), (1, (257,))]:
    q = rng.integers(0, 256, shape, dtype=np.uint8)
    out = ref.relu(ref.u8(q, 0.05, zp))
    assert out.zero_point() == zp
    r_cases.append(dict(q=q, par=np.array([zp], np.float64), out=out.numpy().copy()))
save("ref_relu.npz", r_cases)

# ---- a8 max_pool2d<u8> (src/functional.cc:36-64) ----------------------------
p_cases = []
for k, s, shape in [(3, 2, (2, 5, 55, 55)), (3, 2, (3, 4, 27, 27)), (3, 2, (2, 6, 13, 13)),
                    (2, 2, (2, 20, 24, 24)), (2, 2, (3, 7, 8, 8)), (2, 1, (1, 1, 4, 4)),
                    (1, 2, (1, 1, 4, 4)), (2, 2, (1, 3, 7, 9)), (3, 3, (2, 2, 10, 11)),
                    (5, 1, (1, 2, 5, 5))]:
    q = rng.integers(0, 256, shape, dtype=np.uint8)
    out = ref.max_pool2d(ref.u8(q, 0.05, 77), k, s)
    assert out.zero_point() == 77
    p_cases.append(dict(q=q, par=np.array([k, s], np.float64), out=out.numpy().copy()))
save("ref_maxpool.npz", p_cases)


# ---- calibrator: Calibrator::sample (deterministic while the 1000-slot reservoir fills) + get_range -------
# src/calibrator.cc:6-40.  1000 values, fed in one or several sample() calls, then get_range(quantile).
crng = np.random.default_rng(777)
c_cases = []


def add_c(values, quantile, cuts=()):
    v = np.ascontiguousarray(values, np.float32)
    assert v.size == 1000
    chunks = np.split(v, list(cuts)) if cuts else [v]
    scale, zp = ref.calib_range([np.ascontiguousarray(c) for c in chunks], quantile)
    c_cases.append(dict(x=v, par=np.array([quantile] + list(cuts), np.float64), scale=np.float32(scale),
                        zp=np.int32(zp)))


for qt in (1.0, 0.999, 0.99, 0.9):
    add_c(crng.normal(0.3, 2.0, 1000), qt)
    add_c(crng.uniform(-4, 1, 1000), qt, cuts=(400,))
add_c(crng.uniform(0.5, 9, 1000), 1.0)                    # all positive: zero point 0, scale = max / 255
add_c(crng.uniform(0.5, 9, 1000), 0.99, cuts=(1, 999))
add_c(-crng.uniform(0.5, 9, 1000), 1.0)                   # all negative: zero point 254 / 255
add_c(-crng.uniform(0.5, 9, 1000), 0.95)
add_c(np.zeros(1000), 1.0)                                # nothing but zeros: scale falls back to 1
add_c(np.concatenate([np.zeros(999), [1e-30]]), 1.0)
add_c(np.concatenate([crng.normal(0, 1, 999), [1e6]]), 1.0)   # one outlier stretches the range ...
add_c(np.concatenate([crng.normal(0, 1, 999), [1e6]]), 0.999)  # ... a quantile drops it
add_c(crng.normal(0, 1e-4, 1000), 0.999)
add_c(np.linspace(-1, 1, 1000), 0.5)
save("ref_calib_range.npz", c_cases)
