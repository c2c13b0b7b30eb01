#!/bin/bash
# Round 6, evidence pass on the final code (run through gpurun from the repo root; TAG = output name under gpurun_out/):
#   GPU tests + smoke; rocprofv3 kernel stats + the separate --pmc passes of the default bench command (tools/profile.sh), their summary
#   placed where bench.py looks for it and profiles/lk_traffic.json refreshed from them; the bench lines (EuRoC, KITTI); BA iteration /
#   batch / wall-clock figures and the kernel stats of a batched solve; the lock-step sweep of configs[4]; a randomised parity campaign.
# Copy what should be judged from gpurun_out/<TAG>/ into profiles/ afterwards (tools/r6_collect.sh).
TAG=${1:-r6_final}; FUZZ=${2:-1500}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -1 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $O/smoke.txt
# --- rocprofv3: kernel trace + FETCH_SIZE / WRITE_SIZE / SQ_* passes of the bench's timed loop
bash tools/profile.sh $TAG > $O/profile.log 2>&1
ALG=$(python - <<PY
import json
j = json.loads([l for l in open("$ROOT/gpurun_out/prof_$TAG/trace_bench.json") if l.startswith("{")][-1])
print(j["roofline"]["algorithmic_bytes_per_launch"])
PY
)
python tools/summarize_profile.py gpurun_out/prof_$TAG --update-lk-traffic 4096 $ALG > $O/summarize.log 2>&1; tail -2 $O/summarize.log
cp gpurun_out/prof_$TAG/summary.json profiles/${TAG}_rocprof_summary_seqs4096.json
cp gpurun_out/prof_$TAG/summary.json $O/${TAG}_rocprof_summary_seqs4096.json
cp gpurun_out/prof_$TAG/trace_kernel_stats.csv $O/${TAG}_kernel_stats_seqs4096.csv 2>/dev/null
cp profiles/lk_traffic.json $O/lk_traffic.json
# --- the bench lines (they now find counters whose kernel-source hashes match the tree)
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
# the driver's command line (20 timed steps after 5: boost clocks, no thermal steady state)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err; echo "bench (driver args) rc $?"
timeout 600 python bench.py --workload kitti --no-cpu-baseline --no-extras > $O/bench_kitti.json 2> $O/bench_kitti.err; echo "kitti rc $?"
python - <<PY
import json
j = json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1])
c = j["config5"]
print("value", j["value"], "ms/step", j["ms_per_step"], "frac", j["roofline"]["frac"], "traffic", j["roofline"]["traffic"], "pre", j["roofline_pre"]["ms_per_step"], j["roofline_pre"]["kernels_source"])
print("config5 fps", c["fps"], "s", c["seconds_slowest_rank"], "x", c["lockstep_speedup_vs_streams"], "all", c["every_keyframe_optimised"]["fps"], "cpu", j.get("cpu_baseline", {}).get("value"))
k = json.loads([l for l in open("$O/bench_kitti.json") if l.startswith("{")][-1])
print("kitti value", k["value"], "ms/step", k["ms_per_step"])
PY
# --- batched keyframe detection: kernel stats + counters
bash tools/detect_prof.sh ${TAG}_detect 4096 > $O/detect_prof.log 2>&1; cp gpurun_out/prof_${TAG}_detect/summary.json $O/detect_batch_counters.json
# --- BA
timeout 300 python tools/ba_iter_time.py 2>&1 | tail -3 | tee $O/ba_iter_time.txt
for n in 11 4 1; do timeout 300 python tools/ba_batch_time.py $n 20 2>&1 | tail -1; done | tee $O/ba_batch_time.txt
OV2_DEBUG=1 timeout 300 python tools/ba_batch_time.py 11 1 2>&1 | grep "local_ba_batch" | tail -6 | tee -a $O/ba_batch_time.txt
OV2_DEBUG=1 timeout 300 python tools/localba_wall.py 2>&1 | tail -30 > $O/localba_wall.txt
(cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ba_batch -o t -- python $ROOT/tools/ba_batch_time.py 11 10 > /dev/null 2> $O/prof_ba_batch.err)
find $O/prof_ba_batch -name "*kernel_stats.csv" -exec cp {} $O/ba_batch_11_windows_kernel_stats.csv \;
find $O/prof_ba_batch -name "*_kernel_trace.csv" -delete; find $O/prof_ba_batch -name "*.db" -delete
# --- configs[4]
timeout 800 python tools/lockstep_sweep.py 1 $O/lockstep_sweep.json 2>&1 | cut -c1-260 | tail -14
# --- randomised parity campaign
timeout 1500 python tools/fuzz_parity.py $FUZZ 6 > $O/fuzz.log 2>&1; tail -3 $O/fuzz.log
