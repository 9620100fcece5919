"""The every-wave-in-both-roles form of the first-stage kernel (tools/diag/csrc/i8ie_stem_fused.hip, variant 16, 3 x 3 pools):
bit-exact u8 outputs and INT32 accumulators on the geometries of tests/test_gpu_first_layer.py, incl. image parts, several
images per block and one / two / three feature groups.  Run with I8IE_LIB=tools/diag/libi8ie_hip_diag.so."""
import importlib.util
import os

import numpy as np
import pytest

import abi
import synth

pytestmark = pytest.mark.gpu

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
_spec = importlib.util.spec_from_file_location("first_layer_cases", os.path.join(_ROOT, "tests", "test_gpu_first_layer.py"))
first = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(first)


@pytest.fixture(scope="module")
def gpu():
    c = abi.Ctx(0)
    yield c
    c.close()


@pytest.mark.parametrize("geom", [g for g in first.GEOMS if g[4] <= 96])
def test_fused_roles_first_layer_bit_exact(gpu, orc, geom):
    n, c, h, w, kc, k, stride, pad = geom
    abi.ck(abi.lib().i8ie_ctx_set_option(gpu.h, 2, 16))
    try:
        rng = np.random.default_rng(sum(geom))
        x = rng.uniform(-2.2, 2.6, (n, c, h, w)).astype(np.float32)
        x.flat[::97] = rng.uniform(-9, 9, x.flat[::97].shape)
        q_scale, q_zp = np.float32(0.025), 127
        q_in = orc.quantize(x, q_scale, q_zp)
        cs = synth.conv_case(orc, 5 + sum(geom), n, c, h, w, kc, k, stride, pad, s_in=q_scale, zp_in=q_zp)
        want, want_acc = orc.conv2d(q_in, cs["qw"], cs["qb"], stride, pad, q_scale, q_zp, cs["s_w"], cs["s_out"], cs["zp_out"],
                                    want_acc=True)
        for relu, ob in ((False, 0), (True, 2)):
            names = []
            got, acc = first.run_first(gpu, x, cs["qw"], cs["qb"], stride, pad, q_scale, q_zp, cs["s_w"], cs["s_out"],
                                       cs["zp_out"], relu, ob, (3, 2), names)
            assert "stem_conv_pool" in names, names
            assert np.array_equal(acc, want_acc)
            ref = orc.max_pool2d(orc.relu(want, cs["zp_out"]) if relu else want, 3, 2)
            assert np.array_equal(got, ref), (relu, ob)
    finally:
        abi.ck(abi.lib().i8ie_ctx_set_option(gpu.h, 2, 0))
