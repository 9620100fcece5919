"""Calibrator::sample on the device (include/i8ie_hip.h i8ie_calib_sample_f32; src/calibrator.cc:6-23).

The reference's sampler draws from an unseeded mt19937, so only its algorithm can be pinned, not its stream; the
host-side replay of that stream stays golden-pinned (tests/test_oracle_golden.py, tests/golden/ref_calibrator*).
Here the device path is checked against an independent numpy restatement of the same algorithm with the device's
counter-based draw: every one of the 1000 slots, the count and the resulting range must match exactly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SLOTS = 1000


def _draw(seed, g):
    """splitmix64 of (seed, g) -> uniform in [0, 2000] (csrc/i8ie_elementwise.hip calib_draw)."""
    m = (1 << 64) - 1
    z = (seed + 0x9E3779B97F4A7C15 * (g + 1)) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    z ^= z >> 31
    return ((z >> 32) * 2001) >> 32


def _expected(chunks, seed):
    slots = np.zeros(SLOTS, np.float32)
    g = 0
    for c in chunks:
        for v in c.ravel():
            if g < SLOTS:
                slots[g] = v
            else:
                idx = _draw(seed, g)
                if idx < SLOTS:
                    slots[idx] = v
            g += 1
    return slots, min(g, SLOTS)


def _range(slots, cnt):  # src/calibrator.cc:25-37 with quantile 1
    s = np.sort(slots)
    lo, hi = min(float(s[0]), 0.0), max(float(s[cnt - 1]), 0.0)
    zp = int(np.float32(255 * (0 - np.float32(lo)) / (np.float32(hi) - np.float32(lo) + 1e-09))) & 0xFF
    return lo, hi, zp


@pytest.fixture(scope="module")
def cx():
    import int8inferenceengine_amd  # noqa: F401
    import _CXX_i8ie as cx
    return cx


@pytest.mark.parametrize("sizes,seed", [((10,), 3), ((1000,), 4), ((700, 500), 5), ((5000, 3000, 17), 6), ((40000,), 123456789)])
def test_device_sampler_matches_the_algorithm(cx, sizes, seed):
    rng = np.random.default_rng(seed)
    chunks = [rng.normal(0.3, 2.0, n).astype(np.float32) for n in sizes]
    slots, cnt, scale, zp = cx.calibrator_device_samples(chunks, seed)
    want, want_cnt = _expected(chunks, seed)
    assert cnt == want_cnt
    assert np.array_equal(np.asarray(slots), want)
    lo, hi, want_zp = _range(want, want_cnt)
    assert zp == want_zp and scale > 0


def test_draws_are_uniform_enough():
    d = np.array([_draw(7, g) for g in range(1000, 41000)])
    assert d.min() == 0 and d.max() == 2000
    assert abs((d < SLOTS).mean() - SLOTS / 2001) < 0.01
    hist = np.bincount(d, minlength=2001)
    assert hist.max() < 50 and hist.min() >= 3  # 40000 draws over 2001 values: about 20 each


def test_calibration_on_the_device_gives_a_usable_model(cx):
    """prepare -> one FP32 batch -> convert with the sampler on the device: no layer output is copied to the
    host, the resulting ranges are those of a 1000-value subsample (like the reference's), and the converted
    model agrees with a host-calibrated one on most top-1 decisions."""
    import i8ie
    from int8inferenceengine_amd import workloads as wl
    name = "two_conv"
    sd = wl.synthetic_state_dict(name, seed=5)
    x = wl.synthetic_input(name, 200, seed=8)
    nets = {}
    for mode in ("host", "device"):
        cx.set_calibration_mode(mode)
        try:
            nets[mode] = wl.calibrated(name, sd, calib_seed=11)
        finally:
            cx.set_calibration_mode("auto")
    for a in wl.layer_names(name):
        sh, zh = getattr(nets["host"], a).output_qparams()
        sdv, zd = getattr(nets["device"], a).output_qparams()
        assert 0.5 < sdv / sh < 2.0, (a, sh, sdv)
        assert abs(int(zd) - int(zh)) <= 64
    yh = nets["host"](i8ie.tensor(x)).numpy().argmax(1)
    yd = nets["device"](i8ie.tensor(x)).numpy().argmax(1)
    assert (yh == yd).mean() > 0.8
    assert cx.calibration_mode() == "auto"
