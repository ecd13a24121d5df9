for w in 3 12 20; do python bench.py --workload opt --steps 10 --warmup $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('warmup $w', 'ms/iter', d['ms_per_step'], 'gt', d['ms_gt_render'], 'fwd', d['ms_forward'], 'bwd', d['ms_loss_backward'], 'opt', d['ms_optimiser'])"; done
tools/abn.sh 2 "--pmc off --tf tf1" ab_libs/acfg.so differender_amd/libdifferender_hip.so
