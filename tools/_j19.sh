for w in 2 2 10; do python bench.py --tf tf1 --no-cpu-baseline --pmc off --steps 10 --warmup $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('tf1 warmup $w', 'ms/step', d['ms_per_step'], 'fwd', d['roofline_fwd']['avg_launch_ms'], 'bwd', d['roofline_bwd']['avg_launch_ms'])"; done
python bench.py --tf tf1 --hints off --no-cpu-baseline --pmc off --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('tf1 hints off', 'ms/step', d['ms_per_step'], 'fwd', d['roofline_fwd']['avg_launch_ms'], 'bwd', d['roofline_bwd']['avg_launch_ms'])"
