#!/bin/bash
# A/B timing of two builds of libdifferender_hip.so on the SAME device (devices of the pool differ by ~10 %).
#   usage: tools/ab.sh libA.so libB.so [rounds] [bench args...]
A=$1; B=$2; R=${3:-3}; shift 3
for r in $(seq $R); do for lib in $A $B; do
DIFFERENDER_ALLOW_EXPERIMENT=1 DIFFERENDER_HIP_LIB=$PWD/$lib python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', 'fwd', d['roofline_fwd']['avg_launch_ms'], 'bwd', (d['roofline_bwd'] or {}).get('avg_launch_ms'), 'ms/step', d['ms_per_step'])"
done; done
