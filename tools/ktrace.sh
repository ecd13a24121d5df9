#!/bin/bash
# Per-kernel average durations of one bench.py command (rocprofv3 --kernel-trace --stats), runs ON THE GPU BOX.
#   usage: tools/ktrace.sh OUTNAME [bench args...]      -> gpurun_out/OUTNAME.csv (+ printed)
# (another build: export DIFFERENDER_HIP_LIB=... [DIFFERENDER_ALLOW_EXPERIMENT=1 DIFFERENDER_AB_OLD_ABI=8] before the call)
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/_kt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_kt -- python3 bench.py --no-cpu-baseline --pmc off "$@" > gpurun_out/_kt.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/_kt/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open("gpurun_out/$out.csv", "w") as o:
    w = csv.writer(o); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in [r for i, r in enumerate(rows) if i < 24 or "dr::" in r["Name"]]:   # the top of the list and every kernel of this library
        w.writerow([r["Name"][:120], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
        print("%-100s calls %5s avg_us %9.1f  %5.1f %%" % (r["Name"].replace("void ", "")[:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
grep "^{" gpurun_out/_kt.log | tail -1 > gpurun_out/$out.json
rm -rf gpurun_out/_kt
