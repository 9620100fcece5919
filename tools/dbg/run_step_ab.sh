#!/bin/bash
# the AlexNet step (bench.py) with the product library replaced by A/B builds of one source, in turn (on the GPU box's copy of the tree):
# usage run_step_ab.sh <tag> <file> <name> <name> ...
tag=$1; f=$2; shift; shift
mkdir -p gpurun_out/$tag
cp int8inferenceengine_amd/libi8ie_hip.so /tmp/libi8ie_hip_product.so
for i in 1 2; do
  for n in "$@"; do
    cp "tools/dbg/libi8ie_hip_${f}_$n.so" int8inferenceengine_amd/libi8ie_hip.so
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/$tag/bench_$n.json 2> gpurun_out/$tag/bench_$n.err
    python - <<PY >> gpurun_out/$tag/ab.txt
import json
d=json.loads(open("gpurun_out/$tag/bench_$n.json").read().strip().splitlines()[-1])
k=d["kernel_ms_per_step"]
print("$n", d["value"], d["ms_per_step"], {x:k[x] for x in k if "lin" in x or "stem" in x or "quant" in x})
PY
  done
done
cp /tmp/libi8ie_hip_product.so int8inferenceengine_amd/libi8ie_hip.so
cat gpurun_out/$tag/ab.txt
