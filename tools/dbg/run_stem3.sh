#!/bin/bash
tag=${1:-stem3}
mkdir -p gpurun_out/$tag
timeout -k 10 500 python -m pytest tests/test_gpu_first_layer.py tests/test_gpu_api.py -m gpu -x -q > gpurun_out/$tag/tests.log 2>&1
echo "tests rc $?" | tee -a gpurun_out/$tag/tests.log
tail -3 gpurun_out/$tag/tests.log
I8IE_LIB=tools/diag/libi8ie_hip_diag.so timeout -k 10 500 python -m pytest tools/diag/tests/test_gpu_stem_fused.py -m gpu -x -q > gpurun_out/$tag/tests_diag.log 2>&1
echo "diag tests rc $?" | tee -a gpurun_out/$tag/tests_diag.log
tail -3 gpurun_out/$tag/tests_diag.log
I8IE_LIB=tools/diag/libi8ie_hip_diag.so I8IE_STEM_VARIANTS=0,16,0,16,0,16 timeout -k 10 200 python tools/bench_stem.py 30 1000 > gpurun_out/$tag/bench_stem_diag.txt 2>&1
cat gpurun_out/$tag/bench_stem_diag.txt
I8IE_STEM_VARIANTS=0,0,0 timeout -k 10 200 python tools/bench_stem.py 30 1000 > gpurun_out/$tag/bench_stem.txt 2>&1
cat gpurun_out/$tag/bench_stem.txt
