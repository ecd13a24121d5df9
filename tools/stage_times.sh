#!/bin/bash
# kernel durations of the flat F1/B1 kernels for ablation builds (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
export DIFFERENDER_HIP_LIB=$PWD/$lib
rm -rf gpurun_out/stg
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stg -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/stg.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/stg/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "brick_flat_kernel" in r["Name"] and "true>" not in r["Name"].split("(")[0][-12:]:
        print("$lib", r["Name"].split("(")[0][-52:], "%.3f ms" % (float(r["AverageNs"])/1e6))
PY
done
