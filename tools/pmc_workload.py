#!/usr/bin/env python3
"""Minimal workload for rocprofv3 --pmc passes (counter collection serialises every dispatch, so the full
bench.py is too long under it): calibrate AlexNet once, then a few INT8 forward passes at the bench batch.
usage: rocprofv3 --pmc <counters> -d <dir> -- python3 tools/pmc_workload.py [batch] [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    import torch  # noqa: F401  (HIP runtime first)
    import int8inferenceengine_amd  # noqa: F401
    import i8ie
    from int8inferenceengine_amd import workloads as wl

    net = wl.calibrated("alexnet", wl.synthetic_state_dict("alexnet", seed=42))
    print("calibrated", flush=True)
    x = i8ie.tensor(wl.synthetic_input("alexnet", batch, seed=1234)).prefetch()
    for i in range(steps):
        y = net(x).numpy()
        print("step", i, float(y.sum()), flush=True)


if __name__ == "__main__":
    main()
