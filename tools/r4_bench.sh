#!/bin/bash
# bench line on the GPU box (+ optional extra arguments), printed compactly
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r4b; mkdir -p $O; cd $ROOT
t0=$(date +%s); timeout 900 python bench.py "$@" > $O/bench.json 2> $O/bench.err; echo "bench rc $? $(( $(date +%s) - t0 )) s"; tail -3 $O/bench.err
python - <<'PY'
import json, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r4b")
j = json.loads([l for l in open(os.path.join(O, "bench.json")) if l.startswith("{")][-1])
print("value", j["value"], "ms/step", j["ms_per_step"], "frac", j["roofline"]["frac"])
c5 = j.get("config5") or {}
print("config5", {k: c5.get(k) for k in ("fps", "frames", "seconds_slowest_rank", "concurrent_sequences_per_gpu", "sum_of_single_stream_seconds_per_rank", "device_per_rank", "ba_iters_per_s", "error")})
print("parity", json.dumps(j.get("parity"))[:1500])
print("config2_stream", json.dumps(j.get("config2_stream"))[:600])
PY
