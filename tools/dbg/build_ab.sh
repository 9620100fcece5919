#!/bin/bash
# A/B builds of one product source for same-box comparisons: tools/dbg/libi8ie_hip_<file>_<name>.so = the product objects with
# csrc/<file>.hip taken from a git revision, or from the working tree ("wt"), or from the working tree under a define ("wt-D<MACRO>=<v>").
# usage: tools/dbg/build_ab.sh i8ie_mlin HEAD wt wt-DML_STAGES=4      then  I8IE_LIB=tools/dbg/libi8ie_hip_i8ie_mlin_wt.so python tools/bench_linear.py ...
set -e
cd "$(dirname "$0")/../.."
P=int8inferenceengine_amd
f=$1; shift
for rev in "$@"; do
  d=/tmp/ab_${f}_$rev
  mkdir -p $d
  extra=""
  case "$rev" in
    wt) cp $P/csrc/$f.hip $d/$f.hip ;;
    wt-D*) cp $P/csrc/$f.hip $d/$f.hip; extra="${rev#wt}" ;;
    *) git show $rev:$P/csrc/$f.hip > $d/$f.hip ;;
  esac
  hipcc $extra --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Iinclude -I$P/csrc -c $d/$f.hip -o $d/$f.o
  objs=$(ls $P/build/*.o | grep -v "/$f.o")
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dbg/libi8ie_hip_${f}_$rev.so $objs $d/$f.o
  echo "built $f $rev"
done
