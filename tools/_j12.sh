tools/abn.sh 3 "--pmc off" ab_libs/cur.so ab_libs/empty.so
tools/abn.sh 2 "--pmc off --tf tf1" ab_libs/cur.so ab_libs/empty.so
tools/abn.sh 2 "--pmc off --tf tf1 --scene ct" ab_libs/cur.so ab_libs/empty.so
tools/abn.sh 2 "--pmc off --vol 256 --img 256 --grads none --steps 20" ab_libs/cur.so ab_libs/empty.so
tools/abn_opt.sh 2 ab_libs/cur.so ab_libs/empty.so
for lib in cur empty; do DIFFERENDER_HIP_LIB=$PWD/ab_libs/$lib.so python bench.py --workload opt --scene ct --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib ct-demo', 'ms/iter', d['ms_per_step'], 'gt', d['ms_gt_render'], 'fwd', d['ms_forward'], 'bwd', d['ms_loss_backward'])"; done
