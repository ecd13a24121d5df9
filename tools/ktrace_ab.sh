#!/bin/bash
# Per-kernel durations of several library builds on ONE device, same bench command (runs ON THE GPU BOX):
#   usage: tools/ktrace_ab.sh TAG "lib1 lib2 ..." [bench args...]   -> gpurun_out/TAG_<libname>.csv
tag=$1; libs=$2; shift 2
export DIFFERENDER_ALLOW_EXPERIMENT=1 DIFFERENDER_AB_OLD_ABI=${DIFFERENDER_AB_OLD_ABI:-8}
for lib in $libs; do
  n=$(basename $lib .so); echo "==== $lib"
  DIFFERENDER_HIP_LIB=$PWD/$lib tools/ktrace.sh ${tag}_$n "$@" | grep "dr::" | cut -c1-150
done
