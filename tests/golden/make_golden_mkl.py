#!/usr/bin/env python3
"""Golden vectors for the INT8 contraction from the reference's own GEMM provider.

The reference's contraction lives in Intel MKL (`cblas_gemm_s8u8s32`, called at src/conv2d.cc:131-133 and
src/fully_connected.cc:39-41; the reference pins MKL 2019.5, this image ships the shared objects of MKL 2021.4
in /opt/conda/lib but not mkl.h, so the reference's translation units that include it cannot be built here).
This script does NOT build anything of the reference: it calls that MKL entry point directly through ctypes
with the reference's argument pattern

    cblas_gemm_s8u8s32(CblasRowMajor, CblasNoTrans, CblasTrans, CblasRowOffset,
                       M, N, K, alpha = 1, A (u8, lda = K), ao = 0, B (s8, ldb = K), bo = 0,
                       beta = 0, C (s32, ldc = N), oc)

on seeded operands and stores inputs + outputs as a small .npz fixture (data only).  Generated on an Intel
VNNI host, where this MKL's integer arithmetic is exact; the fixture is what travels.

usage:  MKL_THREADING_LAYER=GNU python tests/golden/make_golden_mkl.py
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("MKL_THREADING_LAYER", "GNU")
mkl = C.CDLL(os.environ.get("I8IE_MKL_RT", "/opt/conda/lib/libmkl_rt.so.1"), mode=C.RTLD_GLOBAL)
ROW_MAJOR, NO_TRANS, TRANS, ROW_OFFSET = 101, 111, 112, 171


def gemm(A, B, oc):
    """A u8 [M, K], B s8 [N, K], oc s32 [N] -> C s32 [M, N], exactly as the reference calls it."""
    M, K = A.shape
    N = B.shape[0]
    Cm = np.empty((M, N), np.int32)
    mkl.cblas_gemm_s8u8s32(ROW_MAJOR, NO_TRANS, TRANS, ROW_OFFSET, C.c_int(M), C.c_int(N), C.c_int(K), C.c_float(1.0),
                           A.ctypes.data_as(C.c_void_p), C.c_int(K), C.c_int8(0), B.ctypes.data_as(C.c_void_p),
                           C.c_int(K), C.c_int8(0), C.c_float(0.0), Cm.ctypes.data_as(C.c_void_p), C.c_int(N),
                           oc.ctypes.data_as(C.c_void_p))
    return Cm


def main():
    rng = np.random.default_rng(424242)
    cases = []

    def add(M, K, N, mode="random"):
        A = rng.integers(0, 256, (M, K), dtype=np.uint8)
        B = rng.integers(-128, 128, (N, K), dtype=np.int8)
        oc = rng.integers(-(1 << 20), 1 << 20, N).astype(np.int32)
        if mode == "extreme":  # 255 x -128 / +127 everywhere: the largest magnitudes the operands allow
            A[...] = 255
            B[::2] = 127
            B[1::2] = -128
        if mode == "zero_oc":
            oc[...] = 0
        cases.append(dict(A=A, B=B, oc=oc, C=gemm(A, B, oc)))

    add(1, 1, 1)
    add(3, 5, 2)
    add(33, 784, 10)            # Linear, MNIST
    add(100, 363, 96)           # AlexNet conv1 rows (K = 3 * 11 * 11)
    add(48, 2304, 96)           # AlexNet conv3 depth
    add(5, 9216, 16)            # fc1 depth
    add(64, 363, 7, "extreme")
    add(16, 4096, 10, "extreme")
    add(40, 1000, 33, "zero_oc")
    add(65, 130, 67)
    flat = {"n_cases": np.asarray(len(cases))}
    for i, c in enumerate(cases):
        for k, v in c.items():
            flat["%d_%s" % (i, k)] = v
    buf = C.create_string_buffer(256)
    mkl.mkl_get_version_string(buf, 256)
    flat["provider"] = np.asarray(buf.value.decode().strip())
    np.savez_compressed(os.path.join(HERE, "mkl_gemm_s8u8s32.npz"), **flat)
    print("mkl_gemm_s8u8s32.npz", len(cases), "cases from", buf.value.decode().strip())


if __name__ == "__main__":
    main()
