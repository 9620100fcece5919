mkdir -p gpurun_out/$1
python -m pytest tests -m gpu -q > gpurun_out/$1/tests.log 2>&1; echo "tests rc $?" >> gpurun_out/$1/tests.log; tail -6 gpurun_out/$1/tests.log
timeout -k 10 300 python tools/calib_time.py device > gpurun_out/$1/calib_time.txt 2>&1; tail -6 gpurun_out/$1/calib_time.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/$1/bench.json 2> gpurun_out/$1/bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/$1/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["fp32_engine_path"]["images_per_sec"], d["parity"]); print(d["kernel_ms_per_step"])
PY
