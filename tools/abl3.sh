#!/bin/bash
for lib in "$@"; do for tf in bench tf1; do
DIFFERENDER_HIP_LIB=$PWD/$lib python bench.py --steps 5 --warmup 2 --no-cpu-baseline --tf $tf 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib tf=$tf', 'value', d['value'], 'fwd', d['roofline_fwd']['avg_launch_ms'], 'bwd', (d['roofline_bwd'] or {}).get('avg_launch_ms'), 'steps/launch', d['roofline_fwd']['voxel_steps_per_launch'])"
done; done
