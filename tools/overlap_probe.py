#!/usr/bin/env python3
"""Do an MFMA-bound conv launch and an HBM-bound streaming launch overlap when issued on two streams?
Two i8ie contexts (one stream each) in one process: conv2 of AlexNet at batch 1000 on ctx A, quantize f32->u8 of
150 M elements on ctx B; serial time vs concurrent time."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
import abi  # noqa: E402


def main():
    lib = abi.lib()
    a, b = abi.Ctx(0), abi.Ctx(0)
    rng = np.random.default_rng(0)
    n, c, h, w, kc, k, stride, pad = 1000, 96, 27, 27, 256, 5, 1, 2
    qw = rng.integers(-63, 64, (kc, c, k, k)).astype(np.int8)
    qb = rng.integers(-63, 64, kc).astype(np.int8)
    L = C.c_void_p()
    abi.ck(lib.i8ie_conv2d_create(a.h, qw.ctypes.data_as(C.c_void_p), qb.ctypes.data_as(C.c_void_p), kc, c, k, k, stride,
                                  pad, C.c_float(0.002), C.byref(L)))
    abi.ck(lib.i8ie_layer_set_output_qparams(L, C.c_float(0.05), C.c_uint8(100)))
    x = rng.integers(0, 256, (n, h + 2 * pad, w + 2 * pad, c), dtype=np.uint8)
    di, out = a.put(x), a.empty((n, h, w, kc), np.uint8)
    m = 150_000_000
    f = b.put(rng.uniform(-1, 1, m).astype(np.float32))
    q = b.empty((m,), np.uint8)

    def conv(reps):
        for _ in range(reps):
            abi.ck(lib.i8ie_layer_forward_fused(L, di.ptr, 1, pad, n, h, w, C.c_float(0.025), C.c_uint8(127), 1, out.ptr,
                                                1, 0, None))

    def stream(reps):
        for _ in range(reps):
            abi.ck(lib.i8ie_quantize_f32_u8(b.h, f.ptr, q.ptr, C.c_int64(m), C.c_float(0.01), C.c_uint8(128)))

    def timed(fn):
        a.sync(); b.sync()
        t0 = time.perf_counter()
        fn()
        a.sync(); b.sync()
        return (time.perf_counter() - t0) * 1e3

    conv(5); stream(5)
    R = 20
    t_conv = timed(lambda: conv(R))
    t_str = timed(lambda: stream(R))

    def both():
        for _ in range(R):
            conv(1)
            stream(1)
    t_both = timed(both)
    print("conv x%d: %.2f ms | stream x%d: %.2f ms (%.2f TB/s) | interleaved on two streams: %.2f ms (serial sum %.2f)"
          % (R, t_conv, R, t_str, 5.0 * m * R / (t_str * 1e-3) / 1e12, t_both, t_conv + t_str))


if __name__ == "__main__":
    main()
