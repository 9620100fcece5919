mkdir -p gpurun_out/r4j
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4j/bench_v0.json 2> gpurun_out/r4j/bench_v0.err
I8IE_KERNEL_VARIANT=50 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4j/bench_v50.json 2> gpurun_out/r4j/bench_v50.err
python - <<PY
import json
for v in ("v0","v50"):
    d=json.loads(open("gpurun_out/r4j/bench_%s.json"%v).read().strip().splitlines()[-1])
    print(v, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"]); print(d["kernel_ms_per_step"])
PY
