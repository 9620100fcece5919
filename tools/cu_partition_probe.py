#!/usr/bin/env python3
"""Does the first stage of batch i + 1 (quantize + conv1 + pool: HBM- and issue-bound, well below the power cap) fit BESIDE conv2-5 of
batch i (at the power cap) when each side gets its own compute units?  Two ctxs on streams with disjoint CU masks
(hipExtStreamCreateWithCUMask, i8ie_ctx_create_on_stream, I8IE_OPT_CU_LIMIT), the same work first one after the other on the whole
chip, then side by side.
usage: python tools/cu_partition_probe.py [conv-side CUs, comma list] [steps]      e.g.  208,192,176 20
$I8IE_PROBE_PER_XCD=1: the conv side's CUs are the first cus/8 of every group of 32 mask bits instead of the first `cus` bits."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import abi  # noqa: E402

hip = C.CDLL("libamdhip64.so")


def hck(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: hip error %d" % (what, rc))


def masked_stream(bits, total=256):
    """stream that may use the CUs in `bits` (bit i of the mask = CU i as the runtime numbers them)"""
    words = (total + 31) // 32
    mask = (C.c_uint32 * words)()
    for i in bits:
        mask[i // 32] |= 1 << (i % 32)
    s = C.c_void_p()
    hck(hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(words), mask), "hipExtStreamCreateWithCUMask")
    return s


class Side:
    """a ctx (own stream, or a given one) with the first-stage layer and conv2 + pool on it"""

    def __init__(self, stream=None, cus=0):
        lib = abi.lib()
        self.lib = lib
        self.g = abi.Ctx.__new__(abi.Ctx)
        self.g.h = C.c_void_p()
        if stream is None:
            abi.ck(lib.i8ie_ctx_create(0, C.byref(self.g.h)))
        else:
            abi.ck(lib.i8ie_ctx_create_on_stream(0, stream, C.byref(self.g.h)))
        if cus:
            abi.ck(lib.i8ie_ctx_set_option(self.g.h, 4, cus))
        rng = np.random.default_rng(0)
        n = 1000
        self.n = n
        # first stage
        qw = rng.integers(-63, 64, (96, 3, 11, 11)).astype(np.int8)
        qb = rng.integers(-63, 64, 96).astype(np.int8)
        self.L1 = C.c_void_p()
        abi.ck(lib.i8ie_conv2d_create(self.g.h, qw.ctypes.data_as(C.c_void_p), qb.ctypes.data_as(C.c_void_p), 96, 3, 11, 11, 4, 2,
                                      C.c_float(0.002), C.byref(self.L1)))
        abi.ck(lib.i8ie_layer_set_output_qparams(self.L1, C.c_float(0.05), C.c_uint8(100)))
        x = rng.uniform(-2.0, 2.4, (n, 3, 224, 224)).astype(np.float32)
        self.dx = self.g.put(x)
        self.o1 = self.g.empty((n, 27 + 4, 27 + 4, 96), np.uint8)
        # conv2 + pool
        qw2 = rng.integers(-63, 64, (256, 96, 5, 5)).astype(np.int8)
        qb2 = rng.integers(-63, 64, 256).astype(np.int8)
        self.L2 = C.c_void_p()
        abi.ck(lib.i8ie_conv2d_create(self.g.h, qw2.ctypes.data_as(C.c_void_p), qb2.ctypes.data_as(C.c_void_p), 256, 96, 5, 5, 1, 2,
                                      C.c_float(0.002), C.byref(self.L2)))
        abi.ck(lib.i8ie_layer_set_output_qparams(self.L2, C.c_float(0.05), C.c_uint8(100)))
        x2 = rng.integers(0, 256, (n, 31, 31, 96), dtype=np.uint8)
        self.di2 = self.g.put(x2)
        self.o2 = self.g.empty((n, 13, 13, 256), np.uint8)

    def stage1(self):
        abi.ck(self.lib.i8ie_layer_forward_f32_input_pool(self.L1, self.dx.ptr, self.n, 224, 224, C.c_float(0.025), C.c_uint8(127), 1, 3, 2,
                                                          self.o1.ptr, 1, 2, None))

    def conv2(self):
        abi.ck(self.lib.i8ie_layer_forward_pool(self.L2, self.di2.ptr, 1, 2, self.n, 27, 27, C.c_float(0.025), C.c_uint8(127), 1, 3, 2,
                                                self.o2.ptr, 1, 0, None))

    def sync(self):
        self.g.sync()


def main():
    splits = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "208,192").split(",")]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    # per AlexNet step: one first stage (0.31 ms) and conv2-5 + fc (0.97 ms = 2.45 x conv2 + pool): 5 conv2 launches per 2 steps
    conv_per_2 = 5
    whole = Side()
    for _ in range(3):
        whole.stage1()
        whole.conv2()
    whole.sync()

    def timed(fn):
        t0 = time.perf_counter()
        fn()
        return (time.perf_counter() - t0) * 1e3

    def seq():
        for s in range(steps):
            whole.stage1()
            for _ in range(conv_per_2 // 2 + (s & 1) * (conv_per_2 & 1)):
                whole.conv2()
        whole.sync()

    def only1():
        for _ in range(steps):
            whole.stage1()
        whole.sync()

    def only2():
        for s in range(steps):
            for _ in range(conv_per_2 // 2 + (s & 1) * (conv_per_2 & 1)):
                whole.conv2()
        whole.sync()

    for rep in range(2):
        print("whole chip, one stream: first stages alone %.3f ms, conv2 launches alone %.3f ms, interleaved in order %.3f ms (%d steps)"
              % (timed(only1), timed(only2), timed(seq), steps), flush=True)
    for cus in splits:
        if os.environ.get("I8IE_PROBE_PER_XCD"):  # the conv side's share taken from every group of 32 mask bits alike
            per = cus // 8
            abits = [32 * x + i for x in range(8) for i in range(per)]
        else:
            abits = list(range(cus))
        bbits = [i for i in range(256) if i not in set(abits)]
        a = Side(masked_stream(abits), cus)        # conv side
        b = Side(masked_stream(bbits), 256 - cus)  # first-stage side
        for _ in range(3):
            a.conv2()
            b.stage1()
        a.sync(); b.sync()

        def side_by_side():
            for s in range(steps):
                b.stage1()
                for _ in range(conv_per_2 // 2 + (s & 1) * (conv_per_2 & 1)):
                    a.conv2()
            a.sync(); b.sync()

        def a_alone():
            for s in range(steps):
                for _ in range(conv_per_2 // 2 + (s & 1) * (conv_per_2 & 1)):
                    a.conv2()
            a.sync()

        def b_alone():
            for _ in range(steps):
                b.stage1()
            b.sync()

        for rep in range(2):
            print("conv side %d CUs / first-stage side %d CUs: conv alone %.3f ms, first stage alone %.3f ms, side by side %.3f ms"
                  % (cus, 256 - cus, timed(a_alone), timed(b_alone), timed(side_by_side)), flush=True)


if __name__ == "__main__":
    main()
