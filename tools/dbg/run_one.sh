#!/bin/bash
# one pytest selection on the GPU: usage run_one.sh <tag> <pytest args...>
tag=$1; shift
mkdir -p gpurun_out/$tag
timeout -k 10 600 python -m pytest "$@" -m gpu -x -q > gpurun_out/$tag/tests.log 2>&1
echo "tests rc $?" | tee -a gpurun_out/$tag/tests.log
tail -15 gpurun_out/$tag/tests.log
