mkdir -p gpurun_out/r4_mlin500
for b in 500 384; do
  python bench.py --batch $b --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r4_mlin500/b${b}_v0.json 2>>gpurun_out/r4_mlin500/err.txt
  I8IE_KERNEL_VARIANT=83 python bench.py --batch $b --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r4_mlin500/b${b}_v83.json 2>>gpurun_out/r4_mlin500/err.txt
done
python - <<PY
import json
for b in (500,384):
    for v in ("v0","v83"):
        d=json.loads(open("gpurun_out/r4_mlin500/b%d_%s.json"%(b,v)).read().strip().splitlines()[-1])
        k=d["kernel_ms_per_step"]
        print(b, v, d["value"], d["ms_per_step"], {x:k[x] for x in k if "lin" in x or "splitk" in x})
PY
