#!/bin/bash
# usage: tools/pmc_lib.sh <outdir> <lib.so> <counters...>   -- PMC pass over bench.py (fwd only) with a given library build
out=$1; lib=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export DIFFERENDER_ALLOW_EXPERIMENT=1 DIFFERENDER_HIP_LIB=$PWD/$lib
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/$out -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --pmc off > gpurun_out/$out.log 2>&1
f=$(find gpurun_out/$out -name "*counter_collection.csv" | head -1)
python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$f")))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
first = None
for r in rows:
    if "brick_flat_kernel" not in r["Kernel_Name"]: continue
    k = r["Kernel_Name"].split("(")[0][-45:]
    first = first or r["Counter_Name"]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == first: cnt[k] += 1
for k, d in agg.items():
    print("$lib", k, {a: "%.4g" % (b / max(cnt[k],1)) for a, b in d.items()})
PY
