#!/bin/bash
# last pass of the round on the final code: GPU tests, smoke, BA statistics / timelines / clocks, bench lines (EuRoC, KITTI)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r4c; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -1 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/r4_close_ba.sh 2>&1 | tail -9
timeout 600 python bench.py --workload kitti --no-cpu-baseline > $O/bench_kitti.json 2> $O/bench_kitti.err; echo "kitti rc $?"
python - <<'PY'
import json, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r4c")
j = json.loads([l for l in open(os.path.join(O, "bench_kitti.json")) if l.startswith("{")][-1])
print("kitti value", j["value"], "ms/step", j["ms_per_step"], "frac", j["roofline"]["frac"], "pre", (j.get("roofline_pre") or {}).get("ms_per_step"))
PY
