#!/bin/bash
# copies what tools/r5_final.sh left under gpurun_out/<TAG>/ into profiles/ (tracked) under the names profiles/README.md lists
TAG=${1:-r5_final}; S=gpurun_out/$TAG; P=profiles
cp $S/bench.json $P/${TAG}_bench.json
cp $S/bench_kitti.json $P/${TAG}_kitti_bench.json
cp $S/${TAG}_kernel_stats_seqs4096.csv $S/${TAG}_rocprof_summary_seqs4096.json $P/
cp $S/lk_traffic.json $P/lk_traffic.json
cp $S/ba_iter_time.txt $P/${TAG}_ba_iter_time.txt
cp $S/ba_batch_time.txt $P/${TAG}_ba_batch_time.txt
cp $S/ba_batch_11_windows_kernel_stats.csv $P/${TAG}_ba_batch_11_windows_kernel_stats.csv
cp $S/localba_wall.txt $P/${TAG}_localba_wall.txt
cp $S/lockstep_sweep.json $P/${TAG}_lockstep_sweep.json
cp $S/fuzz.log $P/${TAG}_fuzz.log
cp $S/pytest_gpu.log $P/${TAG}_pytest_gpu.log
for f in r5_lockstep_hwq.json r5_lockstep_priorities.json r5_lockstep_estimator_groups.json; do [ -f gpurun_out/$f ] && cp gpurun_out/$f $P/$f; done
ls $P | grep -v archive
