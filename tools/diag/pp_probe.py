#!/usr/bin/env python3
"""One layer, one kernel variant, a few launches: the workload for rocprofv3 --pmc passes over the pp kernel.
usage: [rocprofv3 --pmc ... --] python3 tools/pp_probe.py <layer> <variant> [iters]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import abi  # noqa: E402
from bench_layer import LAYERS  # noqa: E402


def main():
    name, variant = sys.argv[1], int(sys.argv[2])
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    lib = abi.lib()
    g = abi.Ctx(0)
    rng = np.random.default_rng(0)
    n, c, h, w, kc, k, stride, pad = LAYERS[name]
    oh, ow = (h - k + 2 * pad) // stride + 1, (w - k + 2 * pad) // stride + 1
    qw = rng.integers(-63, 64, (kc, c, k, k)).astype(np.int8)
    qb = rng.integers(-63, 64, kc).astype(np.int8)
    L = C.c_void_p()
    abi.ck(lib.i8ie_conv2d_create(g.h, qw.ctypes.data_as(C.c_void_p), qb.ctypes.data_as(C.c_void_p), kc, c, k, k,
                                  stride, pad, C.c_float(0.002), C.byref(L)))
    abi.ck(lib.i8ie_layer_set_output_qparams(L, C.c_float(0.05), C.c_uint8(100)))
    x = rng.integers(0, 256, (n, h + 2 * pad, w + 2 * pad, c), dtype=np.uint8)
    di = g.put(x)
    out = g.empty((n, oh, ow, kc), np.uint8)
    abi.ck(lib.i8ie_ctx_set_option(g.h, 2, variant))
    for _ in range(iters):
        abi.ck(lib.i8ie_layer_forward_fused(L, di.ptr, 1, pad, n, h, w, C.c_float(0.025), C.c_uint8(127), 1, out.ptr, 1,
                                            0, None))
    g.sync()
    print("done", name, variant, flush=True)


if __name__ == "__main__":
    main()
