#!/usr/bin/env python3
"""Time Linear layers through the C-ABI for several kernel variants.  usage: python tools/bench_linear.py [variants] [iters] [rows]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import abi  # noqa: E402

LAYERS = {"fc6": (9216, 4096), "fc7": (4096, 4096)}


def main():
    variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,11").split(",")]
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    rows = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "125,250").split(",")]
    lib = abi.lib()
    g = abi.Ctx(0)
    rng = np.random.default_rng(0)
    for name, (k, n) in LAYERS.items():
        qw = rng.integers(-63, 64, (n, k)).astype(np.int8)
        qb = rng.integers(-63, 64, n).astype(np.int8)
        L = C.c_void_p()
        abi.ck(lib.i8ie_linear_create(g.h, qw.ctypes.data_as(C.c_void_p), qb.ctypes.data_as(C.c_void_p), n, k, C.c_float(0.002), C.byref(L)))
        abi.ck(lib.i8ie_layer_set_output_qparams(L, C.c_float(0.05), C.c_uint8(100)))
        for m in rows:
            x = rng.integers(0, 256, (m, k), dtype=np.uint8)
            di = g.put(x)
            out = g.empty((m, n), np.uint8)
            res = {}
            for rep in range(3):
                for v in variants:
                    abi.ck(lib.i8ie_ctx_set_option(g.h, 2, v))
                    for _ in range(3):
                        abi.ck(lib.i8ie_layer_forward_fused(L, di.ptr, 0, 0, m, 0, 0, C.c_float(0.025), C.c_uint8(127), 1, out.ptr, 0, 0, None))
                    g.sync()
                    t0 = time.perf_counter()
                    for _ in range(iters):
                        abi.ck(lib.i8ie_layer_forward_fused(L, di.ptr, 0, 0, m, 0, 0, C.c_float(0.025), C.c_uint8(127), 1, out.ptr, 0, 0, None))
                    g.sync()
                    res.setdefault(v, []).append((time.perf_counter() - t0) / iters * 1e6)
            print(name, "m=%d" % m, " ".join("v%d: %.1f us" % (v, min(t)) for v, t in res.items()), flush=True)
            # per-kernel device time (profile hooks) of the default dispatch
            class _E(C.Structure):
                _fields_ = [("name", C.c_char * 64), ("launches", C.c_uint64), ("total_ms", C.c_double), ("total_ops", C.c_double), ("total_bytes", C.c_double)]
            abi.ck(lib.i8ie_ctx_set_option(g.h, 2, 0))
            abi.ck(lib.i8ie_profile_start(g.h, 0))
            for _ in range(iters):
                abi.ck(lib.i8ie_layer_forward_fused(L, di.ptr, 0, 0, m, 0, 0, C.c_float(0.025), C.c_uint8(127), 1, out.ptr, 0, 0, None))
            ents = (_E * 64)(); cnt = C.c_int(0)
            abi.ck(lib.i8ie_profile_stop(g.h, ents, 64, C.byref(cnt)))
            print("    " + " | ".join("%s %.1f us (%.0f TOPS)" % (ents[i].name.decode().split("|")[0], ents[i].total_ms / ents[i].launches * 1e3, ents[i].total_ops / (ents[i].total_ms * 1e-3) / 1e12) for i in range(cnt.value)), flush=True)
            di.free()
        lib.i8ie_layer_destroy(L)


if __name__ == "__main__":
    main()
