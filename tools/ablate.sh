#!/bin/bash
# usage: tools/ablate.sh  -> prints fwd/bwd kernel ms for variants and gradient subsets
for v in 0 2; do for g in vol+tf vol tf; do
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --variant $v --grads $g 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('variant $v grads $g: fwd', d['roofline_fwd']['avg_launch_ms'], 'bwd', d['roofline_bwd']['avg_launch_ms'], 'ms/step', d['ms_per_step'])"
done; done
