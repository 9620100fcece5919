"""Seeded random geometries for the patch-stationary convolution kernels (csrc/i8ie_pconv.hip, i8ie_tconv.hip)
against the oracle, through the C-ABI: channel counts that give 2 / 4 / 6 / 8 / 10 / 16 chunks per tap (the K-chunk
pairing), odd and even output widths (the padded patch rows), strides 1 and 2, kernels 1 / 3 / 5, feature counts
that exercise one, two and three passes and the 384-wide pass, batches around the CU count (units of a whole band or
of a (band, pass) pair), borders wider than the padding, fused ReLU.  Every output byte is compared; the default
dispatch (variant 0) must agree with the forced kernels."""
import os

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


def _cases(n_cases, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n_cases:
        c = int(rng.choice([32, 64, 96, 128, 160, 256]))
        k = int(rng.choice([1, 3, 3, 5]))
        stride = int(rng.choice([1, 1, 1, 2]))
        pad = int(rng.integers(0, k // 2 + 1))
        h = int(rng.integers(10, 31))
        w = int(rng.integers(10, 31))
        oh, ow = (h - k + 2 * pad) // stride + 1, (w - k + 2 * pad) // stride + 1
        if oh < 4 or ow < 4 or oh * ow < 129 or ow > 64:
            continue
        kc = int(rng.choice([192, 256, 320, 384, 512]))
        n = int(rng.choice([64, 100, 130, 257, 300]))
        if n * c * (h + 2 * pad + 2) * (w + 2 * pad + 2) > 200e6 or n * oh * ow * kc > 200e6:
            continue
        extra = int(rng.integers(0, 2))  # input border wider than the padding
        out.append((n, c, h, w, kc, k, stride, pad, extra, bool(rng.integers(0, 2)), int(rng.integers(0, 3))))
    return out


@pytest.mark.parametrize("case", _cases(int(os.environ.get("I8IE_PCONV_FUZZ_CASES", "18")),
                                        int(os.environ.get("I8IE_PCONV_FUZZ_SEED", "20261004"))))
def test_random_geometry(gpu, orc, case):
    n, c, h, w, kc, k, stride, pad, extra, relu, ob = case
    cs = synth.conv_case(orc, 99 + sum(int(v) for v in case), n, c, h, w, kc, k, stride, pad)
    want = orc.relu(cs["out"], cs["zp_out"]) if relu else cs["out"]
    lib = abi.lib()
    for variant in (0, 50, 70):
        abi.ck(lib.i8ie_ctx_set_option(gpu.h, 2, variant))
        try:
            out, acc = gpu.layer_forward_fused("conv", cs["q_in"], cs["qw"], cs["qb"], cs["s_in"], cs["zp_in"], cs["s_w"],
                                               cs["s_out"], cs["zp_out"], stride=stride, pad=pad, in_nhwc=True,
                                               out_nhwc=True, relu=relu, in_border=pad + extra, out_border=ob,
                                               want_acc=True)[:2]
        finally:
            abi.ck(lib.i8ie_ctx_set_option(gpu.h, 2, 0))
        assert np.array_equal(acc, cs["acc"]), (variant, case)  # INT32 pre-requant accumulators (src/conv2d.cc:131-133)
        assert np.array_equal(out, want), (variant, case)
