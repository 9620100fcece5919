#!/bin/bash
# Sample the GPU's reported shader clock and power while the production conv kernel runs on
# (a) uniformly random and (b) constant operands.  usage: tools/clock_probe.sh > out.txt
cd "$(dirname "$0")/.."
probe() {
  local tag=$1; shift
  env "$@" python tools/bench_layer.py 0 8000 conv2 > /tmp/clock_probe_$tag.txt 2>&1 &
  local pid=$!
  sleep 7   # past imports / uploads, into the timed loops (3 x 8000 launches of ~0.47 ms)
  for i in 1 2 3 4 5 6; do
    echo "== $tag sample $i"
    rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|Power|power" | head -6
    sleep 1
  done
  wait $pid
  cat /tmp/clock_probe_$tag.txt
}
echo "== idle"; rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|Power|power" | head -6
probe random I8IE_DUMMY=1
probe constant I8IE_BENCH_CONST=1
