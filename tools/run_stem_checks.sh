mkdir -p gpurun_out/$1
timeout -k 10 300 python -m pytest tests/test_gpu_first_layer.py -x -q > gpurun_out/$1/tests_first.log 2>&1; echo "tests rc $?" >> gpurun_out/$1/tests_first.log; tail -4 gpurun_out/$1/tests_first.log
timeout -k 10 200 python tools/bench_stem.py 20 1000 > gpurun_out/$1/stem.log 2>&1; cat gpurun_out/$1/stem.log
