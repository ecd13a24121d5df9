#!/bin/bash
# Build a variant of libdifferender_hip.so into ab_libs/<name>.so with extra compiler flags (for tools/ab.sh / abn.sh).
#   usage: tools/mkvariant.sh name [extra hipcc flags...]
# The flag list is the Makefile's (make print-common / print-bwdvol): one definition. A variant built with a what-if switch
# (csrc/dr_experiment.h) marks itself: the Python loader refuses it unless DIFFERENDER_ALLOW_EXPERIMENT=1 (tools/abn.sh sets it).
set -e
name=$1; shift
src=differender_amd/csrc; out=ab_libs/obj_$name
mkdir -p $out
# PATCH=closed_experiments (or any tools/patches/NAME.patch): build from a scratch copy of csrc/ with that patch applied -- the switches
# of experiments that are closed (what-if ablations, per-phase clocks, lane / crossing statistics, the D4 bound's debug output) live
# there, not in the shipped sources; the -D flags of tools/README.md work on the patched copy as they always did
if [ -n "$PATCH" ]; then
  tmp=$(mktemp -d); mkdir -p $tmp/differender_amd $tmp/include
  cp -r $src $tmp/differender_amd/csrc; cp include/*.h $tmp/include/
  (cd $tmp/differender_amd/csrc && patch -s -p1 < $OLDPWD/tools/patches/$PATCH.patch)
  src=$tmp/differender_amd/csrc
fi
COMMON="$(make -s -C $src print-common)"
if [ -n "${NO_LICM_FLAG+x}" ]; then COMMON="${COMMON//-mllvm -disable-machine-licm/}"; fi
pids=()
for f in capi ray_setup march_baseline ray_passes march_flat tf_tape epilogue collective; do
  /opt/rocm/bin/hipcc $COMMON "$@" -c $src/$f.hip -o $out/$f.o & pids+=($!)
done
# B1 in its own translation unit, with its own scheduler strategy (Makefile); BWDVOL_SCHED=default builds it like the rest;
# BWDVOL_EXTRA="-D..." adds switches to that translation unit only
if [ "${BWDVOL_SCHED:-iterative-minreg}" = default ]; then SCHED=""; else SCHED="-mllvm -amdgpu-sched-strategy=${BWDVOL_SCHED:-iterative-minreg}"; fi
/opt/rocm/bin/hipcc $COMMON $SCHED "$@" $BWDVOL_EXTRA -c $src/march_flat_bwdvol.hip -o $out/march_flat_bwdvol.o & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab_libs/$name.so $out/*.o -ldl
rm -rf $out
if [ -n "$tmp" ]; then rm -rf $tmp; fi
echo built ab_libs/$name.so
