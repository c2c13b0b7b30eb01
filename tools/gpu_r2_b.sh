#!/bin/bash
# round-2 GPU session: whole GPU test-suite + the bench line the driver will produce
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r2b; mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time
tail -3 $OUT/bench.time; tail -5 $OUT/bench.err
python - <<'PY'
import json,os
root=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
try:
    d=json.loads(open(root+"/gpurun_out/r2b/bench.json").read().strip().splitlines()[-1])
    for k in ("value","ms_per_step","single_sequence_fps","combined_speedup_vs_cpu","combined_speedup_vs_cpu_batch_amortised","tracking_speedup_vs_cpu_single_sequence"):
        print(k, d.get(k))
    print("roofline", {k:d["roofline"][k] for k in ("frac","avg_launch_ms")})
    print("single", json.dumps(d.get("single_sequence"))[:700])
    print("c5", {k:v for k,v in d.get("config5",{}).items() if k not in ("assignment","workload")})
    print("ba", json.dumps(d.get("ba"))[:1800])
    print("parity", d.get("parity")); print("detect", d.get("detect"))
    cb=d.get("cpu_baseline",{}); print("cpu", {k:cb.get(k) for k in ("value","cores","host_cores","slices_ms","threads","build")}, cb.get("ba"))
except Exception as e:
    print("parse failed", e)
PY
