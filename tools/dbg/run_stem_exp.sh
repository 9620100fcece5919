#!/bin/bash
# phase stamps + time of the first-stage kernel for timing builds: usage run_stem_exp.sh <tag> <n> <n> ...  (0 = the diagnostic build itself)
tag=$1; shift
mkdir -p gpurun_out/$tag
for n in "$@"; do
  lib=tools/dbg/libi8ie_hip_diag_stemexp$n.so
  [ "$n" = 0 ] && lib=tools/diag/libi8ie_hip_diag.so
  echo "== exp $n" >> gpurun_out/$tag/exp.txt
  I8IE_STEM_VARIANTS=0 I8IE_STEM_STAMPS=1 I8IE_LIB=$lib timeout -k 10 120 python tools/bench_stem.py 3 1000 2>&1 | tail -n 2 >> gpurun_out/$tag/exp.txt
  I8IE_STEM_VARIANTS=0,0 I8IE_LIB=$lib timeout -k 10 120 python tools/bench_stem.py 30 1000 2>&1 | tail -n 1 >> gpurun_out/$tag/exp.txt
done
cat gpurun_out/$tag/exp.txt
