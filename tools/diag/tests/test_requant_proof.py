"""Host-side proof of the unguarded requantiser estimate (csrc/i8ie_requant.h, used by csrc/i8ie_pp.hip).

i8ie_requant_fit_host() claims: with the constants it returns, sat_u8(rne(max(fma((float)C, ms, bias), lo))) equals
the reference's down_scale (+ relu) for EVERY int32 C.  Checked here without a GPU: the exact sequence against the
oracle (which is pinned on the reference-compiled goldens), and the estimate against the exact sequence on every
accumulator within +-3 of each of the 255 output steps, on the ends of the int32 range and on a million random
values -- for calibration-like random scales and for "round" scales whose accumulators land exactly on rounding
boundaries (the case where the nominal multiplier fails and a float neighbour is needed)."""
import ctypes as C

import numpy as np
import pytest

import abi


def _fit(lib, sa, sb, sc, zp, relu):
    ms, bias = C.c_float(0), C.c_float(0)
    lib.i8ie_requant_fit_host.restype = C.c_int
    ok = lib.i8ie_requant_fit_host(C.c_float(sa), C.c_float(sb), C.c_float(sc), zp, relu, C.byref(ms), C.byref(bias))
    assert ok in (0, 1)
    return ok, ms.value, bias.value


def _eval(lib, sa, sb, sc, zp, relu, ms, bias, acc):
    acc = np.ascontiguousarray(acc, np.int32)
    ex = np.empty(acc.size, np.uint8)
    es = np.empty(acc.size, np.uint8)
    abi.ck(lib.i8ie_requant_eval_host(C.c_float(sa), C.c_float(sb), C.c_float(sc), zp, relu, C.c_float(ms), C.c_float(bias),
                                      acc.ctypes.data_as(C.c_void_p), C.c_int64(acc.size), ex.ctypes.data_as(C.c_void_p),
                                      es.ctypes.data_as(C.c_void_p)))
    return ex, es


CASES = [(0.025, 0.002, 0.05, 100), (0.025, 0.002, 0.05, 0), (0.025, 0.0041382, 0.0731, 131), (0.031, 0.00077, 0.019, 17),
         (1.0, 1.0, 1.0, 0), (0.5, 0.25, 0.125, 128), (0.0123, 0.00345, 2.5, 250)]


@pytest.mark.parametrize("relu", [0, 1])
@pytest.mark.parametrize("case", CASES + list(range(6)))
def test_fitted_estimate_equals_the_exact_sequence(orc, case, relu):
    lib = abi.lib()
    rng = np.random.default_rng(7 + relu + (10 * case if isinstance(case, int) else 1000))
    if isinstance(case, int):
        sa, sb = np.float32(rng.uniform(0.005, 0.1)), np.float32(rng.uniform(0.0005, 0.01))
        sc, zp = np.float32(rng.uniform(0.01, 3.0)), int(rng.integers(0, 256))
    else:
        sa, sb, sc, zp = np.float32(case[0]), np.float32(case[1]), np.float32(case[2]), case[3]
    ok, ms, bias = _fit(lib, sa, sb, sc, zp, relu)
    # accumulators around every step of the exact function: bisect on the oracle-pinned exact sequence
    lo = zp if relu else 0
    pts = [np.array([-2**31, -2**31 + 1, 2**31 - 2, 2**31 - 1, 0, 1, -1], np.int64)]
    probe = np.arange(-2**31, 2**31, 2**31 // 4096, dtype=np.int64)
    ex_probe, _ = _eval(lib, sa, sb, sc, zp, relu, ms, bias, probe)
    for level in range(lo + 1, 256):
        a, b = -2**31, 2**31
        while a < b:
            mid = (a + b) // 2
            if _eval(lib, sa, sb, sc, zp, relu, ms, bias, np.array([mid]))[0][0] >= level:
                b = mid
            else:
                a = mid + 1
        pts.append(np.arange(a - 3, a + 4, dtype=np.int64))
    pts.append(rng.integers(-2**31, 2**31, 200000, dtype=np.int64))
    span = int(min(2**31 - 1, 300.0 * float(sc) / (float(sa) * float(sb))))
    pts.append(rng.integers(-span, span + 1, 800000, dtype=np.int64))
    acc = np.clip(np.concatenate(pts), -2**31, 2**31 - 1).astype(np.int32)
    ex, es = _eval(lib, sa, sb, sc, zp, relu, ms, bias, acc)
    ref = orc.down_scale(acc, sa, sb, sc, zp)
    if relu:
        ref = orc.relu(ref, zp)
    assert np.array_equal(ex, ref)  # the library's exact sequence is the reference's
    if ok:
        assert np.array_equal(es, ex)  # the proven estimate is exact wherever we look


def test_fit_outcomes():
    """Coarse requantisers (few accumulator values per output level) and "round" scales are provable; fine ones
    mostly are not (the reference's four roundings move about a fifth of the 255 steps by one accumulator value,
    which no single fma reproduces), and the kernel then keeps its guarded estimate."""
    lib = abi.lib()
    assert _fit(lib, np.float32(1.0), np.float32(1.0), np.float32(1.0), 0, 0)[0] == 1
    assert _fit(lib, np.float32(0.5), np.float32(0.25), np.float32(0.125), 128, 1)[0] == 1
    assert _fit(lib, np.float32(0.025), np.float32(0.002), np.float32(0.05), 100, 1)[0] == 1
