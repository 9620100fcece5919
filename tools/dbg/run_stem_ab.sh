#!/bin/bash
# same-box A/B of first-stage kernel builds (tools/dbg/build_stem_ab.sh): usage run_stem_ab.sh <tag> <name> <name> ...
tag=$1; shift
mkdir -p gpurun_out/$tag
for i in 1 2 3; do
  for n in "$@"; do
    echo "== $n" >> gpurun_out/$tag/ab.txt
    I8IE_STEM_VARIANTS=0 I8IE_LIB=tools/dbg/libi8ie_hip_stem_$n.so timeout -k 10 120 python tools/bench_stem.py 30 1000 >> gpurun_out/$tag/ab.txt 2>&1
  done
done
cat gpurun_out/$tag/ab.txt
