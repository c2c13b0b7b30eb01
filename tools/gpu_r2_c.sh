#!/bin/bash
# round-2 GPU session C: FETCH_SIZE calibration on gathers, PMC + kernel-trace passes of the bench, full bench line
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r2c; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/calib -o fetch -- $ROOT/tools/bin/fetch_calib > $OUT/calib.json 2> $OUT/calib.err
timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $OUT/calib -o rdreq -- $ROOT/tools/bin/fetch_calib > /dev/null 2> $OUT/calib2.err
cat $OUT/calib.json
python - <<'PY'
import csv,glob,os,collections
root=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for tag in ("fetch","rdreq"):
    for f in glob.glob(root+"/gpurun_out/r2c/calib/**/%s_counter_collection.csv"%tag, recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)): acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k,v in sorted(acc.items()): print(tag, k, v)
PY
cd $ROOT
bash tools/profile.sh r2v1 "--steps 40 --warmup 10 --no-extras --no-cpu-baseline" > $OUT/profile.log 2>&1; tail -4 $OUT/profile.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time
tail -3 $OUT/bench.time
python - <<'PY'
import json,os
root=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
d=json.loads(open(root+"/gpurun_out/r2c/bench.json").read().strip().splitlines()[-1])
cb=d.get("cpu_baseline",{}); print("cpu", {k:cb.get(k) for k in ("value","cores","host_cores","slices_ms","threads","build")}, cb.get("ba"))
for k in ("value","single_sequence_fps","combined_speedup_vs_cpu","tracking_speedup_vs_cpu_single_sequence"): print(k, d.get(k))
PY
