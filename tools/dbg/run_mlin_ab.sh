#!/bin/bash
# mlin tests on the product build, then bench_linear per A/B build: usage run_mlin_ab.sh <tag> <name> <name> ...
tag=$1; shift
mkdir -p gpurun_out/$tag
timeout -k 10 600 python -m pytest tests/test_gpu_mlin.py -m gpu -q -x > gpurun_out/$tag/tests_mlin.log 2>&1; echo "tests rc $?" >> gpurun_out/$tag/tests_mlin.log; tail -3 gpurun_out/$tag/tests_mlin.log
for i in 1 2; do
  for n in "$@"; do
    echo "== $n" >> gpurun_out/$tag/ab.txt
    I8IE_LIB=tools/dbg/libi8ie_hip_i8ie_mlin_$n.so timeout -k 10 200 python tools/bench_linear.py 0 30 1000,500 >> gpurun_out/$tag/ab.txt 2>&1
  done
done
cat gpurun_out/$tag/ab.txt
