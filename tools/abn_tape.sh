#!/bin/bash
# C3 (--grads tf) with and without the per-sample tape, interleaved on the SAME device: tools/abn_tape.sh rounds [bench args]
R=$1; shift
for r in $(seq $R); do for t in "" "--no-tape"; do
python bench.py --grads tf --steps 8 --warmup 3 --no-cpu-baseline --pmc off $t "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('${t:-tape}', 'fwd', d['roofline_fwd']['avg_launch_ms'], 'bwd', d['roofline_bwd']['avg_launch_ms'], 'ms/step', d['ms_per_step'], 'bwd frac', d['roofline_bwd']['frac'])"
done; done
