#!/bin/bash
# first-layer parity tests against an A/B build, then the same-box A/B: usage run_stem_ab_tests.sh <tag> <tested name> <name> <name> ...
tag=$1; shift
mkdir -p gpurun_out/$tag
I8IE_LIB=tools/dbg/libi8ie_hip_stem_$1.so timeout -k 10 500 python -m pytest tests/test_gpu_first_layer.py -m gpu -x -q > gpurun_out/$tag/tests.log 2>&1
echo "tests ($1) rc $?" | tee -a gpurun_out/$tag/tests.log
tail -2 gpurun_out/$tag/tests.log
bash tools/dbg/run_stem_ab.sh $tag "$@"
