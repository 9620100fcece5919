#!/bin/bash
# stem kernel: parity tests of the first layer, then the isolated bench of the two launches (tools/bench_stem.py)
tag=${1:-stem}
mkdir -p gpurun_out/$tag
timeout -k 10 500 python -m pytest tests/test_gpu_first_layer.py tests/test_gpu_api.py -m gpu -x -q > gpurun_out/$tag/tests.log 2>&1
echo "tests rc $?" | tee -a gpurun_out/$tag/tests.log
tail -3 gpurun_out/$tag/tests.log
I8IE_STEM_VARIANTS=0,14,0,14,0,14 timeout -k 10 200 python tools/bench_stem.py 30 1000 > gpurun_out/$tag/bench_stem.txt 2>&1
cat gpurun_out/$tag/bench_stem.txt
