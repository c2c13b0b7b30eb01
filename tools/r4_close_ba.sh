#!/bin/bash
# BA part of tools/r4_close.sh + the bench line (after a BA-only change)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r4c; mkdir -p $O
for c in config4_mono config4_stereo window; do bash tools/ba_kstats.sh $c > $O/ks_$c.log 2>&1; cp $(find gpurun_out/ksba_$c -name "*kernel_stats.csv" | head -1) $O/ba_kernel_stats_$c.csv; head -1 $O/ks_$c.log; done
for c in config4_mono window; do bash tools/ba_timeline.sh $c > $O/tl_$c.log 2>&1; cp gpurun_out/batl_$c/timeline.txt $O/timeline_$c.txt; find gpurun_out/batl_$c -name "*kernel_trace.csv" -delete; done
bash tools/ba_ticks.sh > $O/ba_ticks.log 2>&1; tail -2 $O/ba_ticks.log
python tools/ba_iter_time.py > $O/ba_iter_time.txt 2>&1; tail -3 $O/ba_iter_time.txt
t0=$(date +%s); timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $? $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r4c")
j = json.loads([l for l in open(os.path.join(O, "bench.json")) if l.startswith("{")][-1])
print("value", j["value"], "ms/step", j["ms_per_step"], "frac", j["roofline"]["frac"], "pre", (j.get("roofline_pre") or {}).get("ms_per_step"), "ba us/it", (j.get("ba") or {}).get("us_per_iteration"))
PY
