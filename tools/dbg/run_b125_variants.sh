#!/bin/bash
# the 125-image step under process-wide kernel variants: usage run_b125_variants.sh <tag> <variant> <variant> ...
tag=$1; shift
mkdir -p gpurun_out/$tag
for i in 1 2; do
  for v in "$@"; do
    I8IE_KERNEL_VARIANT=$v python bench.py --batch 125 --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/$tag/b_$v.json 2> gpurun_out/$tag/b_$v.err
    python - <<PY >> gpurun_out/$tag/ab.txt
import json
d=json.loads(open("gpurun_out/$tag/b_$v.json").read().strip().splitlines()[-1])
print("variant $v", d["value"], d["ms_per_step"], d["kernel_ms_per_step"])
PY
  done
done
cat gpurun_out/$tag/ab.txt
