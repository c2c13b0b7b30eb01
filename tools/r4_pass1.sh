#!/bin/bash
# Round-4 first GPU pass: the whole GPU test suite, the EuRoC bench line, configs[2] (KITTI) with its kernel stats, the BA per-kernel
# profiles and the single-camera tracker's kernel stats.  Run through gpurun from the repo root; outputs under gpurun_out/r4a.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/r4a; mkdir -p $O; cd $ROOT
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $? $(( $(date +%s) - t0 )) s"; tail -5 $O/pytest.log
t0=$(date +%s)
timeout 600 python bench.py > $O/bench_euroc.json 2> $O/bench_euroc.err; echo "bench euroc rc $? $(( $(date +%s) - t0 )) s"
t0=$(date +%s)
timeout 300 python bench.py --workload kitti --no-cpu-baseline > $O/bench_kitti.json 2> $O/bench_kitti.err; echo "bench kitti rc $? $(( $(date +%s) - t0 )) s"
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kitti_ks -o t -- python $ROOT/bench.py --workload kitti --steps 40 --warmup 10 --no-extras --no-cpu-baseline > $O/kitti_ks_bench.json 2> $O/kitti_ks.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/track_ks -o t -- python $ROOT/tools/track_latency.py 300 > $O/track_latency.json 2> $O/track_ks.err
cd $ROOT
for c in config4_mono config4_stereo window; do bash tools/ba_kstats.sh $c r4a/ksba_$c; done
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*.db" -delete
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r4a")
for n in ("bench_euroc", "bench_kitti"):
    try:
        j = json.loads([l for l in open(os.path.join(O, n + ".json")) if l.startswith("{")][-1])
        print(n, j["value"], j["ms_per_step"], j.get("roofline", {}).get("frac"), j.get("roofline", {}).get("launch_us"))
    except Exception as e:
        print(n, "unreadable", e)
PY
