#!/bin/bash
# Linear / API tests, then the AlexNet step at shard sizes where the many-row Linear kernel's row tile matters
tag=${1:-ml64c}
mkdir -p gpurun_out/$tag
timeout -k 10 600 python -m pytest tests/test_gpu_mlin.py tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -q -x > gpurun_out/$tag/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/$tag/tests.log; tail -3 gpurun_out/$tag/tests.log
for b in 500 384 1000; do
  python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/$tag/b$b.json 2> gpurun_out/$tag/b$b.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/$tag/b$b.json").read().strip().splitlines()[-1]); print($b, d["value"], d["ms_per_step"], d["kernel_ms_per_step"], d["parity"]["logits_bit_exact_vs_oracle"])
PY
done
