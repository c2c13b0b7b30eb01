#!/bin/bash
# rocprofv3 evidence for the batched keyframe detector (tools/detect_batch_time.py): kernel stats + HBM / SQ counters in their own passes.
# Usage (through gpurun, from the repo root): bash tools/detect_prof.sh <tag> [S]
set -u
TAG=${1:-r6_detect}; S=${2:-4096}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
python $ROOT/tools/detect_batch_time.py $S 3 > $OUT/time.txt 2>&1
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python $ROOT/tools/detect_batch_time.py $S 2 > /dev/null 2> $OUT/trace.err
for pm in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"; do
  n=$(echo $pm | cut -d' ' -f1)
  timeout 500 rocprofv3 --pmc $pm --kernel-trace --output-format csv -d $OUT -o pmc_$n -- python $ROOT/tools/detect_batch_time.py $S 1 > /dev/null 2> $OUT/pmc_$n.err
done
cd $ROOT
python - "$OUT" <<'PY'
import csv, glob, os, sys, json, collections
out = sys.argv[1]
res = {}
f = glob.glob(out + "/**/trace_kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    res["kernel_stats"] = [{k: r[k] for k in r} for r in rows[:12]]
pm = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/pmc_*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        pm[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res["pmc_mean_per_dispatch"] = {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"dispatches": max(len(v) for v in cs.values())} for k, cs in pm.items() if "k_" in k}
res["time"] = open(out + "/time.txt").read().strip().splitlines()[-6:]
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
PY
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*counter_collection.csv" -size +2M -delete
