#!/bin/bash
# Round-4 closing pass on the GPU box: BA timelines of the committed tree, the bench line (EuRoC, then KITTI), the GPU test suite.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r4c; mkdir -p $O
for c in config4_mono window; do bash tools/ba_timeline.sh $c > $O/tl_$c.log 2>&1; cp gpurun_out/batl_$c/timeline.txt $O/timeline_$c.txt; head -1 $O/tl_$c.log; find gpurun_out/batl_$c -name "*kernel_trace.csv" -delete; done
cd $ROOT
t0=$(date +%s); timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $? $(( $(date +%s) - t0 )) s"; tail -3 $O/bench.err
timeout 600 python bench.py --workload kitti --no-cpu-baseline > $O/bench_kitti.json 2> $O/bench_kitti.err; echo "kitti rc $?"
python - <<'PY'
import json, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r4c")
for n in ("bench.json", "bench_kitti.json"):
    j = json.loads([l for l in open(os.path.join(O, n)) if l.startswith("{")][-1])
    print(n, "value", j["value"], "ms/step", j["ms_per_step"], "frac", j["roofline"]["frac"])
    print("  roofline_pre", json.dumps(j.get("roofline_pre"))[:900])
    print("  ba.roofline", json.dumps((j.get("ba") or {}).get("roofline"))[:300])
PY
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
