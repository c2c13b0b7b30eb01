#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r2g; mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_clahe.py tests/test_gpu_frontend.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
timeout 300 python bench.py --steps 30 --warmup 10 --no-extras --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("default value", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "lk_ms/step", round(d["lk_ms_per_step"],3), "frac", round(d["roofline"]["frac"],4))
PY
