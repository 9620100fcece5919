#!/bin/bash
# Timing builds of the first-stage kernel (csrc/i8ie_stem.hip under -DSTEM_EXP=<bits>, see the file; results of bits 2 and 4 are WRONG
# by design): the diagnostic build's objects with that one file recompiled -> tools/dbg/libi8ie_hip_diag_stemexp<n>.so
# usage: python tools/diag/build_diag.py; tools/dbg/build_stem_exp.sh 1 2 3 ...
set -e
cd "$(dirname "$0")/../.."
P=int8inferenceengine_amd
for n in "$@"; do
  (
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Iinclude -I$P/csrc -DSTEM_EXP=$n \
      -DI8IE_DIAG -Itools/diag/include -c $P/csrc/i8ie_stem.hip -o /tmp/stem_exp$n.o
    objs=$(ls tools/diag/build/*.o | grep -v i8ie_stem.o)
    hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dbg/libi8ie_hip_diag_stemexp$n.so $objs /tmp/stem_exp$n.o
    echo "built exp $n"
  ) &
done
wait
