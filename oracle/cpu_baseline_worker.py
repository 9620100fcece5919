#!/usr/bin/env python3
"""Timed CPU baseline with the reference's own GEMM provider (TEST / MEASUREMENT INFRASTRUCTURE ONLY).

bench.py's cpu_baseline leg starts this in a clean process (torch, which bench.py has loaded, carries its own
MKL): the oracle's whole-network forward (oracle/pipeline.py) with its contraction routed through Intel MKL's
cblas_gemm_s8u8s32 -- the routine the reference calls at src/conv2d.cc:131-133 / src/fully_connected.cc:39-41
-- when that runtime is installed and reproduces the exact integer results on this host (oracle/orc.use_mkl).

usage: cpu_baseline_worker.py <in.npz> <out.json> <threads> <seconds>
in.npz: entry_json, x [n,c,h,w] f32, ref_logits [n,classes] f32 (the oracle's own-kernel result),
        per layer <name>__qw, <name>__qb, <name>__sw, <name>__qp (scale, zero point)
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import orc  # noqa: E402
import pipeline  # noqa: E402


def main():
    src, dst, threads, seconds = sys.argv[1], sys.argv[2], int(sys.argv[3]), float(sys.argv[4])
    d = np.load(src, allow_pickle=False)
    layers, spec, shape = json.loads(str(d["entry_json"]))
    entry = ({k: tuple(v) for k, v in layers.items()}, [tuple(op) for op in spec], tuple(shape))
    x, ref = d["x"], d["ref_logits"]
    qlayers = {k: (d[k + "__qw"], d[k + "__qb"], np.float32(d[k + "__sw"])) for k in entry[0]}
    qparams = {k: (np.float32(d[k + "__qp"][0]), int(d[k + "__qp"][1])) for k in entry[0]}
    orc.set_num_threads(threads)
    out = {"provider": orc.use_mkl(), "threads": orc.num_threads()}
    if out["provider"] is not None:
        k = int(min(16, x.shape[0]))  # probe: exactness against the own-kernel logits, and a rate estimate
        t0 = time.perf_counter()
        got = pipeline.forward(entry, x[:k], qlayers, qparams)
        rate = k / (time.perf_counter() - t0)
        out["logits_bit_exact_vs_own_kernel"] = bool(np.array_equal(got.view(np.uint32), ref[:k].view(np.uint32)))
        n = int(max(k, min(x.shape[0], rate * seconds)))  # about `seconds` of work
        passes, t0 = 0, time.perf_counter()
        while True:
            pipeline.forward(entry, x[:n], qlayers, qparams)
            passes += 1
            el = time.perf_counter() - t0
            if passes >= 8 or el * (passes + 1) / passes > seconds:
                break
        out.update(images=n, passes=passes, seconds=round(el, 2), images_per_sec=round(n * passes / el, 2))
    json.dump(out, open(dst, "w"))


if __name__ == "__main__":
    main()
