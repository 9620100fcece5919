#!/bin/bash
# bench.py at 1000 images: eager launches against one HIP-graph replay per batch, alternating
tag=$1
mkdir -p gpurun_out/$tag
for i in 1 2 3; do
  for g in off on; do
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline --graph $g > gpurun_out/$tag/bench_$g.json 2> gpurun_out/$tag/bench_$g.err
    python - <<PY >> gpurun_out/$tag/ab.txt
import json
d=json.loads(open("gpurun_out/$tag/bench_$g.json").read().strip().splitlines()[-1])
print("graph $g", d["value"], d["ms_per_step"], d.get("value_synchronous"))
PY
  done
done
cat gpurun_out/$tag/ab.txt
