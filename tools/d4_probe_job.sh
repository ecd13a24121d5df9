#!/bin/bash
# GPU box: the three dumps of tools/d4_bound_probe.py for one configuration, then the comparison.  usage: tools/d4_probe_job.sh N WH SR TF [MODE]
# needs ab_libs/d4dbg.so (PATCH=closed_experiments tools/mkvariant.sh d4dbg -DDR_D4_DEBUG) and ab_libs/d4off.so (tools/mkvariant.sh d4off -DDR_D4_BUDGET_OVERRIDE=3.0e38f)
set -e
O=gpurun_out/d4probe; mkdir -p $O
T="$1_$2_$3_$4_${5:-0}"
DIFFERENDER_ALLOW_EXPERIMENT=1 DIFFERENDER_HIP_LIB=$PWD/ab_libs/d4dbg.so python tools/d4_bound_probe.py dump $O/b_$T.npz "$@"
DIFFERENDER_ALLOW_EXPERIMENT=1 DIFFERENDER_HIP_LIB=$PWD/ab_libs/d4off.so python tools/d4_bound_probe.py dump $O/f_$T.npz "$@"
python tools/d4_bound_probe.py dump $O/s_$T.npz "$@"
echo "== $T"
python tools/d4_bound_probe.py compare $O/b_$T.npz $O/f_$T.npz $O/s_$T.npz
rm -f $O/*_$T.npz
