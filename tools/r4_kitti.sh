#!/bin/bash
# configs[2] refresh: CLAHE tests, pre-processing A/B, the KITTI bench line and its kernel stats
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r4c; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_clahe.py -x -q 2>&1 | tail -1
for m in 1 2 -1; do python tools/pre_micro.py 4096 8 $m 2>&1 | tail -1; done
( cd /tmp; export TMPDIR=/tmp; rm -rf $O/kitti_ks; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kitti_ks -o t -- python $ROOT/bench.py --workload kitti --steps 40 --warmup 10 --no-extras --no-cpu-baseline > $O/kitti_ks_bench.json 2> $O/kitti_ks.err )
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*.db" -delete
timeout 600 python bench.py --workload kitti --no-cpu-baseline > $O/bench_kitti.json 2> $O/bench_kitti.err; echo "kitti rc $?"
python - <<'PY'
import json, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r4c")
j = json.loads([l for l in open(os.path.join(O, "bench_kitti.json")) if l.startswith("{")][-1])
print("kitti value", j["value"], "ms/step", j["ms_per_step"], "frac", j["roofline"]["frac"], "pre", (j.get("roofline_pre") or {}).get("ms_per_step"))
PY
