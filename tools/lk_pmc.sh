#!/bin/bash
# Stall-attribution counters of k_fb_klt3 on the GPU box (tools/lk_micro.py workload; separate --pmc passes, kernel-trace only).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/lkpmc
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+|TCP_[A-Z_0-9]+|TA_[A-Z_0-9]+" | sort -u > $OUT/avail.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_IFETCH" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_CYCLES_SALU SQ_WAVES SQ_INSTS_VMEM_RD" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT -o set$i -- python $ROOT/tools/lk_micro.py ${1:-2048} > $OUT/set$i.log 2> $OUT/set$i.err
done
cd $ROOT
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_fb_klt3" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for k, v in acc.items():
        v.sort()
        # first 6 dispatches = (levels 3, max_iter 30); dispatches 12..17 = max_iter 0
        a = [x for _, x in v[1:6]]; b = [x for _, x in v[13:18]]
        print("%-44s full %.4g   stage-only %.4g" % (k, sum(a) / len(a), sum(b) / max(len(b), 1)))
PY
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
