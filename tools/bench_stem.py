#!/usr/bin/env python3
"""First stage of AlexNet (FP32 input -> quantize -> conv1 -> relu -> max-pool 3/2) through the C-ABI, per-kernel device
times from the profile hooks: the first-stage kernel (csrc/i8ie_stem.hip, variant 0) against the older chain
(quantize + repack, conv_smallc, max-pool; variant 11).  usage: python tools/bench_stem.py [iters] [n]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import abi  # noqa: E402


class Entry(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("launches", C.c_uint64), ("total_ms", C.c_double),
                ("total_ops", C.c_double), ("total_bytes", C.c_double)]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    lib = abi.lib()
    g = abi.Ctx(0)
    rng = np.random.default_rng(0)
    c, h, w, kc, k, stride, pad = 3, 224, 224, 96, 11, 4, 2
    oh = ow = (h - k + 2 * pad) // stride + 1
    ph = pw = (oh - 3) // 2 + 1
    qw = rng.integers(-63, 64, (kc, c, k, k)).astype(np.int8)
    qb = rng.integers(-63, 64, kc).astype(np.int8)
    L = C.c_void_p()
    abi.ck(lib.i8ie_conv2d_create(g.h, qw.ctypes.data_as(C.c_void_p), qb.ctypes.data_as(C.c_void_p), kc, c, k, k, stride, pad,
                                  C.c_float(0.002), C.byref(L)))
    abi.ck(lib.i8ie_layer_set_output_qparams(L, C.c_float(0.05), C.c_uint8(100)))
    x = rng.uniform(-2.0, 2.4, (n, c, h, w)).astype(np.float32)
    dx = g.put(x)
    out = g.empty((n, ph + 4, pw + 4, kc), np.uint8)
    for variant in [int(v) for v in os.environ.get("I8IE_STEM_VARIANTS", "0,11,0,11").split(",")]:
        abi.ck(lib.i8ie_ctx_set_option(g.h, 2, variant))
        for _ in range(3):
            abi.ck(lib.i8ie_layer_forward_f32_input_pool(L, dx.ptr, n, h, w, C.c_float(0.025), C.c_uint8(127), 1, 3, 2, out.ptr, 1, 2, None))
        g.sync()
        abi.ck(lib.i8ie_profile_start(g.h, 0))
        for _ in range(iters):
            abi.ck(lib.i8ie_layer_forward_f32_input_pool(L, dx.ptr, n, h, w, C.c_float(0.025), C.c_uint8(127), 1, 3, 2, out.ptr, 1, 2, None))
        ents = (Entry * 64)()
        cnt = C.c_int(0)
        abi.ck(lib.i8ie_profile_stop(g.h, ents, 64, C.byref(cnt)))
        tot = 0.0
        parts = []
        for i in range(cnt.value):
            e = ents[i]
            ms = e.total_ms / e.launches * (e.launches / iters)
            tot += ms
            tops = e.total_ops / (e.total_ms * 1e-3) / 1e12 if e.total_ops else 0
            parts.append("%s %.4f ms%s" % (e.name.decode().split("|")[0], ms, (" (%.0f TOPS)" % tops) if tops else (" (%.0f GB/s)" % (e.total_bytes / (e.total_ms * 1e-3) / 1e9))))
        print("variant %d n %d: total %.4f ms | %s" % (variant, n, tot, " | ".join(parts)), flush=True)
    abi.ck(lib.i8ie_ctx_set_option(g.h, 2, 0))
    lib.i8ie_layer_destroy(L)
    g.close()


if __name__ == "__main__":
    main()
