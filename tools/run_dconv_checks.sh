# one GPU session: the deferred-epilogue kernel's tests (diagnostic build: python tools/diag/build_diag.py first), then layer timings
# (variant 50 = i8ie_pconv.hip, 55 = tools/diag/csrc/i8ie_dconv.hip)
mkdir -p gpurun_out/$1
export I8IE_LIB=tools/diag/libi8ie_hip_diag.so
timeout -k 10 900 python -m pytest tools/diag/tests/test_gpu_dconv.py -m gpu -q > gpurun_out/$1/tests_dconv.log 2>&1; echo "tests rc $?" >> gpurun_out/$1/tests_dconv.log; tail -5 gpurun_out/$1/tests_dconv.log
timeout -k 10 300 python tools/bench_layer.py 50,55 20 conv2,conv3,conv4,conv5 > gpurun_out/$1/bench_layer.txt 2>&1
I8IE_BENCH_POOL=3,2 timeout -k 10 300 python tools/bench_layer.py 50,55 20 conv2,conv5 >> gpurun_out/$1/bench_layer.txt 2>&1
cat gpurun_out/$1/bench_layer.txt
