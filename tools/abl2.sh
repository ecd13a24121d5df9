#!/bin/bash
# per-lib: quick parity subset + bench
for lib in "$@"; do
echo "== $lib"
DIFFERENDER_HIP_LIB=$PWD/$lib timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "flat and (forward_parity or backward_parity)" 2>&1 | tail -1
DIFFERENDER_HIP_LIB=$PWD/$lib python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', 'fwd', d['roofline_fwd']['avg_launch_ms'], 'bwd', (d['roofline_bwd'] or {}).get('avg_launch_ms'), 'ms/step', d['ms_per_step'])"
done
