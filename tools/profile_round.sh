#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes over bench.py.
# Writes summaries under gpurun_out/profiles_$TAG/ ; copy the ones to keep into profiles/.
#   usage: tools/profile_round.sh r01 [bench args...]
TAG=${1:-r01}; shift
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BENCH="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --pmc off $@"
echo "== kernel trace =="
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_full.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/kernel_stats_full.csv")))
with open("$OUT/kernel_stats.csv", "w") as f:
    w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in [r for r in rows if "dr::" in r["Name"] or "rocclr" in r["Name"]]:
        w.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
print(open("$OUT/kernel_stats.csv").read())
PY
pass() {  # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- $BENCH > $OUT/$name.log 2>&1
}
echo "== PMC passes =="
pass pmc_fetch FETCH_SIZE
pass pmc_write WRITE_SIZE
pass pmc_tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum
pass pmc_sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
pass pmc_sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
python - <<PY
import csv, collections, glob, json
res = collections.defaultdict(dict)
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(f)):
        if "dr::" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    for k, d in agg.items():
        for c, v in d.items():
            res[k][c] = v / cnt[k][c]   # average per launch
json.dump(res, open("$OUT/pmc_per_launch.json", "w"), indent=1, sort_keys=True)
for k, d in res.items():
    fetch = d.get("FETCH_SIZE", 0) * 1024; write = d.get("WRITE_SIZE", 0) * 1024
    print(k[:70], "| FETCH_SIZE(raw) %.1f MB  x2-corrected %.1f MB  WRITE_SIZE %.1f MB  L2 hit %.3f" % (
        fetch / 1e6, 2 * fetch / 1e6, write / 1e6,
        d.get("TCC_HIT_sum", 0) / max(d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0), 1)))
PY
