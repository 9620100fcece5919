"""Variants that exist in the diagnostic build only: the few-row Linear kernel (csrc/i8ie_skinny.hip, $I8IE_SKINNY),
the two-stage 256-row form of the tiled contraction kernel (variant 7).  Against the oracle, through the C-ABI."""
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


LARGE_GEOMS = [
    (400, 128, 13, 13, 256, 3, 1, 1),
    (400, 64, 13, 13, 384, 3, 1, 1),
    (400, 32, 13, 13, 320, 3, 1, 1),
    (300, 16, 31, 31, 256, 3, 2, 1),
]


@pytest.mark.parametrize("variant", [4, 7, 8, 9])
@pytest.mark.parametrize("geom", LARGE_GEOMS)
def test_tile_shape_experiments_bit_exact(gpu, orc, geom, variant):
    n, c, h, w, kc, k, stride, pad = geom
    cs = synth.conv_case(orc, 77 + sum(geom), n, c, h, w, kc, k, stride, pad)
    lib = abi.lib()
    abi.ck(lib.i8ie_ctx_set_option(gpu.h, 2, variant))
    try:
        out = gpu.layer_forward_fused("conv", cs["q_in"], cs["qw"], cs["qb"], cs["s_in"], cs["zp_in"], cs["s_w"],
                                      cs["s_out"], cs["zp_out"], stride=stride, pad=pad, in_nhwc=True, out_nhwc=True,
                                      relu=True, in_border=pad, out_border=1, want_acc=False)[0]
    finally:
        abi.ck(lib.i8ie_ctx_set_option(gpu.h, 2, 0))
    assert np.array_equal(out, orc.relu(cs["out"], cs["zp_out"]))


def test_opt_in_few_row_linear_kernel(orc):
    """i8ie_skinny.hip (opt-in with $I8IE_SKINNY=1, read once per process): a K slice of the activations resident
    in LDS, fragment-ordered weights streamed into registers.  Same accumulators and outputs as everything else."""
    import os
    import subprocess
    import sys

    code = r'''
import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import torch, abi, synth, orc  # (torch first: one HIP runtime per process)
g = abi.Ctx(0)
for (m, k, n, flat) in ((125, 9216, 300, (256, 6, 6)), (7, 4096, 4096, None), (128, 1040, 129, None), (33, 512, 64, None)):
    cs = synth.linear_case(orc, 500 + m + k + n, m, k, n)
    for relu in (False, True):
        out, acc, _ = g.layer_forward_fused("linear", cs["q_in"], cs["qw"], cs["qb"], cs["s_in"], cs["zp_in"], cs["s_w"],
                                            cs["s_out"], cs["zp_out"], relu=relu, in_nhwc=flat is not None, flat_chw=flat)
        want = orc.relu(cs["out"], cs["zp_out"]) if relu else cs["out"]
        assert np.array_equal(acc, cs["acc"]) and np.array_equal(out, want), (m, k, n, relu)
print("skinny ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, I8IE_SKINNY="1"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "skinny ok" in r.stdout, r.stdout + r.stderr
