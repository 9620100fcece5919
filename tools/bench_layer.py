#!/usr/bin/env python3
"""Time single layers through the C-ABI for several kernel variants, interleaved in one process.
usage: python tools/bench_layer.py [variants comma list] [iters]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import abi  # noqa: E402

LAYERS = {  # name: (n, c, h, w, kc, k, stride, pad)
    "conv1": (1000, 3, 224, 224, 96, 11, 4, 2),
    "conv2": (1000, 96, 27, 27, 256, 5, 1, 2),
    "conv3": (1000, 256, 13, 13, 384, 3, 1, 1),
    "conv4": (1000, 384, 13, 13, 384, 3, 1, 1),
    "conv5": (1000, 384, 13, 13, 256, 3, 1, 1),
    # K sweep at a fixed tile count (per-tile overhead vs k-loop rate): 13 x 13, N = 384
    "c128": (1000, 128, 13, 13, 384, 3, 1, 1),
    "c256": (1000, 256, 13, 13, 384, 3, 1, 1),
    "c512": (1000, 512, 13, 13, 384, 3, 1, 1),
    "c768": (1000, 768, 13, 13, 384, 3, 1, 1),
    "c1024": (1000, 1024, 13, 13, 384, 3, 1, 1),
    # the same with 768 * k tiles exactly (M = 128 * 256 * 3 / 3 rows, N = 384): no partial last round
    "full256": (768, 256, 16, 16, 384, 3, 1, 1),
    "full768": (768, 768, 16, 16, 384, 3, 1, 1),
}


def main():
    variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    names = sys.argv[3].split(",") if len(sys.argv) > 3 else list(LAYERS)
    lib = abi.lib()
    g = abi.Ctx(0)
    rng = np.random.default_rng(0)
    for name in names:
        n, c, h, w, kc, k, stride, pad = LAYERS[name]
        n = int(os.environ.get("I8IE_BENCH_N", n))
        oh, ow = (h - k + 2 * pad) // stride + 1, (w - k + 2 * pad) // stride + 1
        qw = rng.integers(-63, 64, (kc, c, k, k)).astype(np.int8)
        if os.environ.get("I8IE_BENCH_CONST"):  # constant operands: how much of the rate is data-dependent power?
            qw[...] = 1
        qb = rng.integers(-63, 64, kc).astype(np.int8)
        L = C.c_void_p()
        abi.ck(lib.i8ie_conv2d_create(g.h, qw.ctypes.data_as(C.c_void_p), qb.ctypes.data_as(C.c_void_p), kc, c, k, k,
                                      stride, pad, C.c_float(0.002), C.byref(L)))
        abi.ck(lib.i8ie_layer_set_output_qparams(L, C.c_float(0.05), C.c_uint8(100)))
        if c % 16 == 0:
            x = rng.integers(0, 256, (n, h + 2 * pad, w + 2 * pad, c), dtype=np.uint8)
            layout, border = 1, pad
        else:
            x = rng.integers(0, 256, (n, c, h, w), dtype=np.uint8)
            layout, border = 0, 0
        if os.environ.get("I8IE_BENCH_CONST"):
            x[...] = 129
        di = g.put(x)
        out = g.empty((n, oh, ow, kc), np.uint8)
        ops = 2.0 * n * oh * ow * kc * c * k * k
        res = {}
        # I8IE_BENCH_POOL=3,2: the AlexNet layers that are followed by a max-pool (conv2, conv5) run with it folded in
        pool = [int(v) for v in os.environ["I8IE_BENCH_POOL"].split(",")] if os.environ.get("I8IE_BENCH_POOL") and name in ("conv2", "conv5") else None

        def fwd():
            if pool:
                abi.ck(lib.i8ie_layer_forward_pool(L, di.ptr, layout, border, n, h, w, C.c_float(0.025), C.c_uint8(127), 1, pool[0], pool[1],
                                                   out.ptr, 1, 0, None))
            else:
                abi.ck(lib.i8ie_layer_forward_fused(L, di.ptr, layout, border, n, h, w, C.c_float(0.025), C.c_uint8(127), 1, out.ptr, 1, 0, None))

        for rep in range(3):
            for v in variants:
                abi.ck(lib.i8ie_ctx_set_option(g.h, 2, v))
                for _ in range(3):
                    fwd()
                g.sync()
                t0 = time.perf_counter()
                for _ in range(iters):
                    fwd()
                g.sync()
                ms = (time.perf_counter() - t0) / iters * 1e3
                res.setdefault(v, []).append(ms)
        print(name + ("+pool" if pool else ""), " ".join("v%d: %.4f ms (%.0f TOPS)" % (v, min(t), ops / (min(t) * 1e-3) / 1e12) for v, t in res.items()),
              flush=True)
        lib.i8ie_layer_destroy(L)
        di.free()
        out.free()
    g.close()


if __name__ == "__main__":
    main()
