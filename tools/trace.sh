#!/bin/bash
# usage: tools/trace.sh lib.so [bench args] -> per-kernel avg ms of dr:: kernels
lib=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/trace_tmp
DIFFERENDER_ALLOW_EXPERIMENT=1 DIFFERENDER_HIP_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_tmp -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --pmc off "$@" > gpurun_out/trace_tmp.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/trace_tmp/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "dr::" in r["Name"]: print("$lib", r["Name"].split("(")[0][-62:], "calls", r["Calls"], "avg_us %.1f" % (float(r["AverageNs"])/1e3), "total_ms %.2f" % (float(r["TotalDurationNs"])/1e6))
PY
tail -1 gpurun_out/trace_tmp.log | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('rays individually', d['rays_marched_individually'], 'fwd', d['roofline_fwd']['avg_launch_ms'])
except Exception as e: print('no json', e)"
