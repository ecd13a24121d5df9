#!/bin/bash
# Build a variant of libdifferender_hip.so into ab_libs/<name>.so with extra compiler flags (for tools/ab.sh).
#   usage: tools/mkvariant.sh name [extra hipcc flags...]
set -e
name=$1; shift
src=differender_amd/csrc; out=ab_libs/obj_$name
mkdir -p $out
COMMON="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -munsafe-fp-atomics -ffp-contract=off -fno-slp-vectorize ${NO_LICM_FLAG--mllvm -disable-machine-licm}"
pids=()
for f in capi ray_setup march_baseline ray_passes march_flat epilogue collective; do
  /opt/rocm/bin/hipcc $COMMON "$@" -c $src/$f.hip -o $out/$f.o & pids+=($!)
done
# B1 in its own translation unit, with its own scheduler strategy (Makefile); BWDVOL_SCHED=default builds it like the rest;
# BWDVOL_EXTRA="-D..." adds switches to that translation unit only
if [ "${BWDVOL_SCHED:-iterative-minreg}" = default ]; then SCHED=""; else SCHED="-mllvm -amdgpu-sched-strategy=${BWDVOL_SCHED:-iterative-minreg}"; fi
/opt/rocm/bin/hipcc $COMMON $SCHED "$@" $BWDVOL_EXTRA -c $src/march_flat_bwdvol.hip -o $out/march_flat_bwdvol.o & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab_libs/$name.so $out/*.o -ldl
rm -rf $out
echo built ab_libs/$name.so
