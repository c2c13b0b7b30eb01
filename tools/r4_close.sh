#!/bin/bash
# Round-4 closing pass on the GPU box (run through gpurun): GPU test suite, rocprofv3 stats + PMC passes of the bench step
# (tools/profile.sh), the KITTI step's kernel stats, the bench lines (EuRoC, KITTI), BA per-kernel statistics / timelines / phase
# clocks, a fuzz campaign.  Everything lands under gpurun_out/r4c/ (and gpurun_out/prof_<tag>/).
TAG=${1:-r4v2}; FUZZ=${2:-3000}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r4c; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
bash tools/profile.sh $TAG > $O/profile.log 2>&1; tail -2 $O/profile.log
( cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kitti_ks -o t -- python $ROOT/bench.py --workload kitti --steps 40 --warmup 10 --no-extras --no-cpu-baseline > $O/kitti_ks_bench.json 2> $O/kitti_ks.err )
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*.db" -delete
for c in config4_mono config4_stereo window; do bash tools/ba_kstats.sh $c > $O/ks_$c.log 2>&1; cp $(find gpurun_out/ksba_$c -name "*kernel_stats.csv" | head -1) $O/ba_kernel_stats_$c.csv; head -1 $O/ks_$c.log; done
for c in config4_mono window; do bash tools/ba_timeline.sh $c > $O/tl_$c.log 2>&1; cp gpurun_out/batl_$c/timeline.txt $O/timeline_$c.txt; find gpurun_out/batl_$c -name "*kernel_trace.csv" -delete; done
bash tools/ba_ticks.sh > $O/ba_ticks.log 2>&1; tail -2 $O/ba_ticks.log
t0=$(date +%s); timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $? $(( $(date +%s) - t0 )) s"; tail -2 $O/bench.err
timeout 600 python bench.py --workload kitti --no-cpu-baseline > $O/bench_kitti.json 2> $O/bench_kitti.err; echo "kitti rc $?"
python - <<'PY'
import json, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r4c")
for n in ("bench.json", "bench_kitti.json"):
    j = json.loads([l for l in open(os.path.join(O, n)) if l.startswith("{")][-1])
    print(n, "value", j["value"], "ms/step", j["ms_per_step"], "frac", j["roofline"]["frac"], "pre", (j.get("roofline_pre") or {}).get("ms_per_step"), "ba us/it", (j.get("ba") or {}).get("us_per_iteration"))
PY
timeout 900 python tools/fuzz_parity.py $FUZZ 4242 > $O/fuzz.log 2>&1; tail -2 $O/fuzz.log
