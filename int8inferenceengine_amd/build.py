"""In-tree build of the native pieces (gfx950 only).

    python -m int8inferenceengine_amd.build          # build what is stale
    python -m int8inferenceengine_amd.build --force

Produces, next to the sources (git-ignored, but shipped to the GPU box):
    int8inferenceengine_amd/libi8ie_hip.so                     HIP kernels + C-ABI (include/i8ie_hip.h)
    int8inferenceengine_amd/_CXX_i8ie.cpython-*.so             pybind11 module over the C-ABI
hipcc cross-compiles for gfx950 without a GPU present.
"""
import os
import subprocess
import sys
import sysconfig

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INC = os.path.join(ROOT, "include")
OBJ = os.path.join(PKG, "build")

HIP_SOURCES = ["i8ie_ctx.hip", "i8ie_elementwise.hip", "i8ie_gemm.hip", "i8ie_igemm.hip", "i8ie_pconv.hip", "i8ie_tconv.hip", "i8ie_first.hip", "i8ie_stem.hip", "i8ie_flin.hip", "i8ie_mlin.hip", "i8ie_layer.hip", "i8ie_fp32.hip"]
# -ffp-contract=off: the fp32 epilogue must round exactly like the reference's
# SSE2 build (no FMA contraction); IEEE divide/sqrt is hipcc's default and is kept.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
             "-Wall", "-Wno-unused-function", "-I" + INC, "-I" + CSRC]

LIB = os.path.join(PKG, "libi8ie_hip.so")
EXT = os.path.join(PKG, "_CXX_i8ie" + sysconfig.get_config_var("EXT_SUFFIX"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _headers():
    hs = [os.path.join(INC, "i8ie_hip.h")]
    hs += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    return hs


def build_hip(force=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "hipcc")
    objs, jobs = [], []
    hdrs = _headers()
    for src in HIP_SOURCES:
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + hdrs):
            jobs.append([hipcc] + HIP_FLAGS + ["-c", s, "-o", o])
        objs.append(o)
    if jobs:  # translation units are independent: compile them side by side (bounded: the build box has 8 CPUs)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=int(os.environ.get("I8IE_BUILD_JOBS", "6"))) as ex:
            list(ex.map(_run, jobs))
    if force or _stale(LIB, objs):
        _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


def build_ext(force=False):
    import pybind11

    src = os.path.join(CSRC, "pybind_module.cc")
    if not os.path.exists(src):
        return None
    if force or _stale(EXT, [src, LIB] + _headers()):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-ffp-contract=off",
               "-I" + INC, "-I" + CSRC, "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"],
               src, "-o", EXT, "-L" + PKG, "-li8ie_hip", "-Wl,-rpath,$ORIGIN"]
        _run(cmd)
    return EXT


def build_all(force=False):
    build_hip(force)
    build_ext(force)


if __name__ == "__main__":
    build_all("--force" in sys.argv)
