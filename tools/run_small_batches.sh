# the per-GPU shards of the 2 / 4 / 8-GPU configurations on one GPU: step time and per-kernel breakdown
mkdir -p gpurun_out/$1
for b in 125 250 500; do
  python bench.py --batch $b --steps 50 --warmup 5 --no-cpu-baseline --graph off > gpurun_out/$1/bench_b$b.json 2> gpurun_out/$1/bench_b$b.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/$1/bench_b$b.json").read().strip().splitlines()[-1])
print($b, d["value"], d["ms_per_step"]); print(d["kernel_ms_per_step"])
PY
done
python bench.py --batch 125 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/$1/bench_b125_graph.json 2> gpurun_out/$1/bench_b125_graph.err
python -c "
import json
d=json.loads(open('gpurun_out/$1/bench_b125_graph.json').read().strip().splitlines()[-1])
print('125 graph', d['value'], d['ms_per_step'], d.get('launch_mode'))"
