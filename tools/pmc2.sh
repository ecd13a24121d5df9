#!/bin/bash
# usage: tools/pmc2.sh <outdir> "<bench args>" <counters...>
out=$1; bargs=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/$out -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --pmc off $bargs > gpurun_out/$out.log 2>&1
f=$(find gpurun_out/$out -name "*counter_collection.csv" | head -1)
python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$f")))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
first = None
for r in rows:
    if "dr::" not in r["Kernel_Name"]: continue
    k = r["Kernel_Name"].split("(")[0][-50:]
    first = first or r["Counter_Name"]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == first: cnt[k] += 1
for k, d in agg.items():
    print(k, cnt[k], {a: "%.4g" % (b / max(cnt[k],1)) for a, b in d.items()})
PY
