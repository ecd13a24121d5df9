#!/bin/bash
# Time N builds of the library on the SAME device, interleaved over R rounds.
#   usage: tools/abn.sh rounds "bench args" libA.so libB.so ...
R=$1; ARGS=$2; shift 2
for r in $(seq $R); do for lib in "$@"; do
DIFFERENDER_ALLOW_EXPERIMENT=1 DIFFERENDER_AB_OLD_ABI=${DIFFERENDER_AB_OLD_ABI:-8} DIFFERENDER_HIP_LIB=$PWD/$lib python bench.py --steps 5 --warmup 2 --no-cpu-baseline $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', 'fwd', d['roofline_fwd']['avg_launch_ms'], 'bwd', (d['roofline_bwd'] or {}).get('avg_launch_ms'), 'ms/step', d['ms_per_step'], 'repaired', d.get('rays_repaired'), 'individually', d.get('rays_marched_individually'))"
done; done
