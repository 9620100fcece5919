#!/bin/bash
# Timing builds of the deferred-epilogue kernel: libi8ie_hip.so with i8ie_dconv.hip compiled under -DDC_EXP=<n> (see the file), the
# other objects taken from the diagnostic build (python tools/diag/build_diag.py first).  usage: tools/dbg/build_dconv_exp.sh 1 3 5 ...  ->  tools/dbg/libi8ie_hip_exp<n>.so
# (I8IE_LIB=tools/dbg/libi8ie_hip_exp<n>.so python tools/bench_layer.py ...).  Results of these builds are WRONG by design.
set -e
cd "$(dirname "$0")/../.."
P=int8inferenceengine_amd
for n in "$@"; do
  (
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Iinclude -I$P/csrc -DDC_EXP=$n \
      -DI8IE_DIAG -Itools/diag/include -c tools/diag/csrc/i8ie_dconv.hip -o /tmp/dconv_exp$n.o
    objs=$(ls tools/diag/build/*.o | grep -v i8ie_dconv.o)
    hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dbg/libi8ie_hip_exp$n.so $objs /tmp/dconv_exp$n.o
    echo "built exp $n"
  ) &
done
wait
