#!/bin/bash
# first-layer tests against the first named A/B build of i8ie_stem, then the in-step A/B: usage run_stem_step_ab.sh <tag> <tested> <name> ...
tag=$1; shift
mkdir -p gpurun_out/$tag
I8IE_LIB=tools/dbg/libi8ie_hip_i8ie_stem_$1.so timeout -k 10 500 python -m pytest tests/test_gpu_first_layer.py -m gpu -x -q > gpurun_out/$tag/tests.log 2>&1
echo "tests ($1) rc $?" | tee -a gpurun_out/$tag/tests.log
tail -2 gpurun_out/$tag/tests.log
bash tools/dbg/run_step_ab.sh $tag i8ie_stem "$@"
