mkdir -p gpurun_out/$1
for lib in tools/dbg/libi8ie_hip_exp*.so; do
  echo "== $lib" >> gpurun_out/$1/exp.txt
  I8IE_LIB=$lib timeout -k 10 200 python tools/bench_layer.py 50,55 20 conv2,conv5 >> gpurun_out/$1/exp.txt 2>&1
  I8IE_DCONV_STAMPS=1 I8IE_LIB=$lib timeout -k 10 200 python tools/bench_layer.py 55 2 conv2 2>&1 | grep -E "wave 3" | tail -1 | cut -c1-250 >> gpurun_out/$1/exp.txt
done
cat gpurun_out/$1/exp.txt
