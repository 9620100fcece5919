"""Which kernels the mid-size random networks of tests/test_gpu_fuzz.py reach (profile hooks around one forward each).
usage: python tools/midsize_fuzz_coverage.py"""
import sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import int8inferenceengine_amd
import i8ie, _CXX_i8ie as cx
from int8inferenceengine_amd import workloads as wl
import test_gpu_fuzz as tf
tot = collections.Counter()
for case in range(40):
    rng = np.random.default_rng(90_000 + case)
    entry = tf._random_midsize_network(rng)
    name = "_cov_%d" % case
    wl.NETWORKS[name] = entry
    sd = wl.synthetic_state_dict(name, seed=91_000 + case)
    net = wl.calibrated(name, sd, calib_batch=wl.synthetic_input(name, 16, seed=case))
    batch = int(rng.choice([200, 260, 330]))
    x = i8ie.tensor(wl.synthetic_input(name, batch, seed=92_000 + case))
    net(x).numpy()
    cx.profile_start(mfma_only=False)
    net(x).numpy()
    ents = cx.profile_stop()
    names = sorted(set(str(k).split("|")[0] for k in ents))
    for n in names: tot[n] += 1
    print(case, batch, [(k, v[1:]) for k, v in entry[0].items() if v[0] == "conv"], [s for s in entry[1] if s[0] != "layer"][:6], names, flush=True)
print(tot)
