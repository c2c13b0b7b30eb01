#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r2f; mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_clahe.py tests/test_gpu_frontend.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
for t in 0 1; do
  OV2_LK3_TILED=$t timeout 300 python bench.py --steps 30 --warmup 10 --no-extras --no-cpu-baseline > $OUT/bench_tiled$t.json 2> $OUT/bench_tiled$t.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_tiled$t.json").read().strip().splitlines()[-1])
print("LK3_TILED=$t value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "lk_ms/step", round(d["lk_ms_per_step"],3), "frac", round(d["roofline"]["frac"],4), "tracked", d["tracked_fraction"])
PY
done
