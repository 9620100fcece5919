#!/bin/bash
# A/B builds of the first-stage kernel for same-box comparisons: tools/dbg/libi8ie_hip_stem_<name>.so = the product objects with
# i8ie_stem.hip taken from a git revision (or the working tree: "wt").  usage: tools/dbg/build_stem_ab.sh HEAD wt
# (wt-D<MACRO>=<v>: the working tree compiled under that define)
# then  I8IE_LIB=tools/dbg/libi8ie_hip_stem_HEAD.so python tools/bench_stem.py 3 1000
set -e
cd "$(dirname "$0")/../.."
P=int8inferenceengine_amd
for rev in "$@"; do
  src=/tmp/stem_ab_$rev/i8ie_stem.hip
  mkdir -p /tmp/stem_ab_$rev
  extra=""
  case "$rev" in
    wt) cp $P/csrc/i8ie_stem.hip $src ;;
    wt-D*) cp $P/csrc/i8ie_stem.hip $src; extra="${rev#wt}" ;;   # e.g. wt-DSTEM_DMA_VEC=0: the working tree under that define
    *) git show $rev:$P/csrc/i8ie_stem.hip > $src ;;
  esac
  hipcc $extra --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Iinclude -I$P/csrc -c $src -o /tmp/stem_ab_$rev/i8ie_stem.o
  objs=$(ls $P/build/*.o | grep -v i8ie_stem.o)
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dbg/libi8ie_hip_stem_$rev.so $objs /tmp/stem_ab_$rev/i8ie_stem.o
  echo "built $rev"
done
