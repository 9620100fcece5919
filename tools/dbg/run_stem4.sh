#!/bin/bash
tag=${1:-stem4}
mkdir -p gpurun_out/$tag
timeout -k 10 500 python -m pytest tests/test_gpu_first_layer.py tests/test_gpu_api.py -m gpu -x -q > gpurun_out/$tag/tests.log 2>&1
echo "tests rc $?" | tee -a gpurun_out/$tag/tests.log
tail -3 gpurun_out/$tag/tests.log
shift
bash tools/dbg/run_stem_ab.sh $tag "$@"
