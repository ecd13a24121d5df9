#!/bin/bash
# Register / scratch / occupancy summary of the kernels in one translation unit (compile only, no GPU needed).
#   usage: tools/kres.sh march_flat [extra hipcc flags...]
F=$1; shift
cd "$(dirname "$0")/../differender_amd/csrc"
# (the flag list is the Makefile's; for B1's own unit: tools/kres.sh march_flat_bwdvol -mllvm -amdgpu-sched-strategy=iterative-minreg)
/opt/rocm/bin/hipcc $(make -s print-common) -Wno-everything "$@" \
  -Rpass-analysis=kernel-resource-usage -c $F.hip -o /dev/null 2>&1 | python3 -c "
import re, sys
cur = {}
for line in sys.stdin:
    m = re.search(r'remark: ([A-Za-z ]+(?:\[bytes/lane\]|\[waves/SIMD\]|\[bytes/block\])?): (\S+)', line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2)
    if k == 'Function Name':
        cur = {'name': v}
    cur[k] = v
    if k.startswith('LDS Size'):
        import subprocess
        name = subprocess.run(['c++filt', cur['name']], capture_output=True, text=True).stdout.strip()
        name = re.sub(r'\(dr::BrickParams.*', '', name).replace('void dr::', '')
        print('%-78s vgpr %3s sgpr %3s scratch %3s occ %s' % (name[:78], cur.get('VGPRs'), cur.get('SGPRs'), cur.get('ScratchSize [bytes/lane]'), cur.get('Occupancy [waves/SIMD]')))
"
