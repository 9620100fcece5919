mkdir -p gpurun_out/$1
python -m pytest tests -m gpu -x -q > gpurun_out/$1/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/$1/tests.log; tail -4 gpurun_out/$1/tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/$1/bench.json 2> gpurun_out/$1/bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/$1/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"]); print(d["kernel_ms_per_step"])
PY
tail -2 gpurun_out/$1/bench.err
