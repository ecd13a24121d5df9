#!/bin/bash
# Interleaved timing of library builds on the OPT demo workload (same device): tools/abn_opt.sh rounds lib...
# (MSE loss only -- bench.py's default too; --dssim adds the reference's DSSIM term; the JSON line names the loss that was timed)
R=$1; shift
for r in $(seq $R); do for lib in "$@"; do
DIFFERENDER_ALLOW_EXPERIMENT=1 DIFFERENDER_AB_OLD_ABI=${DIFFERENDER_AB_OLD_ABI:-8} DIFFERENDER_HIP_LIB=$PWD/$lib python bench.py --workload opt --no-dssim --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', 'ms/iter', d['ms_per_step'], 'gt', d['ms_gt_render'], 'fwd', d['ms_forward'], 'bwd', d['ms_loss_backward'])"
done; done
