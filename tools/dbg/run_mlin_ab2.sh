#!/bin/bash
# bench_linear per A/B build of i8ie_mlin (no tests): usage run_mlin_ab2.sh <tag> <name> <name> ...
tag=$1; shift
mkdir -p gpurun_out/$tag
for i in 1 2; do
  for n in "$@"; do
    echo "== $n" >> gpurun_out/$tag/ab.txt
    I8IE_LIB=tools/dbg/libi8ie_hip_i8ie_mlin_$n.so timeout -k 10 200 python tools/bench_linear.py 0 30 1000 2>&1 | grep -v mlin_128 >> gpurun_out/$tag/ab.txt
  done
done
cat gpurun_out/$tag/ab.txt
