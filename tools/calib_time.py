import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import int8inferenceengine_amd, i8ie, _CXX_i8ie as cx
from int8inferenceengine_amd import workloads as wl
sd = wl.synthetic_state_dict("alexnet", seed=42)
x = wl.synthetic_input("alexnet", 100, seed=7)
mode = sys.argv[1] if len(sys.argv) > 1 else "device"
cx.set_calibration_mode(mode)
print("calibration sampler:", mode)
for rep in range(3):
    net = wl.build("alexnet"); net.load(sd)
    xt = i8ie.tensor(x).prefetch(); cx.synchronize()
    t0 = time.perf_counter(); y = net(xt); y.numpy(); cx.synchronize(); t_fp32 = time.perf_counter() - t0
    t0 = time.perf_counter(); net.prepare(); t_prep = time.perf_counter() - t0
    cx.profile_start(False)
    t0 = time.perf_counter(); net(xt).numpy(); cx.synchronize(); t_cal = time.perf_counter() - t0
    prof = cx.profile_stop()
    t0 = time.perf_counter(); net.convert(); cx.synchronize(); t_conv = time.perf_counter() - t0
    print("rep %d: fp32 forward %.1f ms | prepare %.1f ms | calibration forward %.1f ms | convert %.1f ms" % (rep, t_fp32*1e3, t_prep*1e3, t_cal*1e3, t_conv*1e3))
    if rep == 2:
        for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])[:8]:
            print("   %-28s launches %3d  %.2f ms" % (k, v[0], v[1]))
