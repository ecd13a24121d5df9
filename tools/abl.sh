#!/bin/bash
for lib in "$@"; do
DIFFERENDER_HIP_LIB=$PWD/$lib python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', 'fwd', d['roofline_fwd']['avg_launch_ms'], 'bwd', (d['roofline_bwd'] or {}).get('avg_launch_ms'), 'ms/step', d['ms_per_step'])"
done
