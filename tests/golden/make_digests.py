#!/usr/bin/env python3
"""Writes tests/golden/alexnet_digests.json: SHA-256 digests of the ORACLE's per-layer u8 outputs and logits for
the bench's AlexNet workload (weights seed 42, input seed 1234 / 5), at the output qparams the product's seeded
calibration produces.  Needs a GPU only for that calibration (the qparams are then part of the fixture, so the
CPU test can replay the oracle alone).  Guards against the oracle and the kernels drifting together between
rounds.

usage (GPU box):  python tests/golden/make_digests.py gpurun_out/alexnet_digests.json
"""
import hashlib
import json
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def f32_hex(x):
    return struct.pack("<f", float(np.float32(x))).hex()


def digests(entry, x, qlayers, qparams):
    import pipeline

    cap = {}
    logits = pipeline.forward(entry, x, qlayers, qparams, capture=cap)
    d = {k: hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest() for k, v in cap.items()
         if isinstance(v, np.ndarray)}
    d["_logits_f32"] = hashlib.sha256(np.ascontiguousarray(logits).tobytes()).hexdigest()
    return d


def main():
    import pipeline
    from int8inferenceengine_amd import workloads as wl

    name = "alexnet"
    sd = wl.synthetic_state_dict(name)
    net = wl.calibrated(name, sd)
    qparams = {a: getattr(net, a).output_qparams() for a in wl.layer_names(name)}
    entry = wl.NETWORKS[name]
    qlayers = pipeline.quantize_layers(entry, sd)
    out = {"network": name, "weights_seed": 42, "calibration": "workloads.calibrated defaults (batch 32, seed 99, sampler seed 7)",
           "qparams": {a: {"scale_f32_hex": f32_hex(s), "scale": float(np.float32(s)), "zero_point": int(z)}
                       for a, (s, z) in qparams.items()},
           "cases": []}
    for batch, seed in ((4, 5), (100, 5), (1000, 1234)):
        x = wl.synthetic_input(name, batch, seed=seed)
        out["cases"].append({"batch": batch, "input_seed": seed, "sha256": digests(entry, x, qlayers, qparams)})
    json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
    print("wrote", sys.argv[1])


if __name__ == "__main__":
    main()
