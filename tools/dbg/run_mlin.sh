mkdir -p gpurun_out/$1
timeout -k 10 600 python -m pytest tests/test_gpu_mlin.py -m gpu -q > gpurun_out/$1/tests_mlin.log 2>&1; echo "tests rc $?" >> gpurun_out/$1/tests_mlin.log; tail -5 gpurun_out/$1/tests_mlin.log
timeout -k 10 300 python tools/bench_linear.py 0,11 30 1000,500,384 > gpurun_out/$1/bench_linear.txt 2>&1; cat gpurun_out/$1/bench_linear.txt
