#!/usr/bin/env python3
"""Diagnostic build of the C-ABI library: the product sources compiled with -DI8IE_DIAG plus the experiment kernels
kept here (csrc/i8ie_pp.hip: persistent ping-pong contraction; csrc/i8ie_skinny.hip: few-row Linear; csrc/i8ie_lgemm.hip:
many-row Linear with weights straight from L2, variant 82; csrc/i8ie_dconv.hip: the deferred-epilogue convolution of round 4,
variant 55), giving
tools/diag/libi8ie_hip_diag.so.  Nothing in the package, bench.py or tests/ loads it; point $I8IE_LIB at it:

    python tools/diag/build_diag.py
    I8IE_LIB=tools/diag/libi8ie_hip_diag.so python -m pytest tools/diag/tests -m gpu -q

What the flag adds: in-kernel phase stamps (variants 51, 71-74), the tile-shape / staging experiments of the tiled
contraction kernel (variants 4, 6-10), the ping-pong kernel (20-49), timing experiments of the two-team kernel
(72-75, $I8IE_TCONV_SPLIT), weights fetched per pass (53), $I8IE_SKINNY.  tools/README.md lists them."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from int8inferenceengine_amd import build as b  # noqa: E402

OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libi8ie_hip_diag.so")


def main():
    os.makedirs(OBJ, exist_ok=True)
    srcs = [os.path.join(b.CSRC, s) for s in b.HIP_SOURCES]
    srcs += [os.path.join(HERE, "csrc", s) for s in ("i8ie_pp.hip", "i8ie_skinny.hip", "i8ie_lgemm.hip", "i8ie_dconv.hip", "i8ie_stem_fused.hip")]
    jobs, objs = [], []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s).replace(".hip", ".o"))
        objs.append(o)
        jobs.append(["hipcc"] + b.HIP_FLAGS + ["-DI8IE_DIAG", "-I" + os.path.join(HERE, "include"), "-c", s, "-o", o])
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=int(os.environ.get("I8IE_BUILD_JOBS", "6"))) as ex:
        list(ex.map(subprocess.check_call, jobs))
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    print(LIB)


if __name__ == "__main__":
    main()
