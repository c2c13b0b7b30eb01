#!/bin/bash
# Issue / stall counters of the pre-processing kernels (tools/pre_micro.py workload; separate --pmc passes, kernel-trace only).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prepmc
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_IFETCH SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT -o set$i -- python $ROOT/tools/pre_micro.py ${1:-4096} 4 > $OUT/set$i.log 2> $OUT/set$i.err
done
cd $ROOT
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-28:]
        if "clahe" in k or "pyr" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        for c, v in d.items(): print("%-30s %-24s %.4g" % (k, c, sum(v[1:]) / max(len(v) - 1, 1)))
PY
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
