#!/bin/bash
# copies what tools/r6_final.sh left under gpurun_out/<TAG>/ into profiles/ (tracked) under the names profiles/README.md lists
TAG=${1:-r6_final}; S=gpurun_out/$TAG; P=profiles
cp $S/bench.json $P/${TAG}_bench.json
cp $S/bench_kitti.json $P/${TAG}_kitti_bench.json
cp $S/bench_driver_args.json $P/${TAG}_bench_driver_args.json
cp $S/detect_batch_counters.json $P/${TAG}_detect_batch_counters.json
cp $S/${TAG}_kernel_stats_seqs4096.csv $S/${TAG}_rocprof_summary_seqs4096.json $P/
cp $S/lk_traffic.json $P/lk_traffic.json
cp $S/ba_iter_time.txt $P/${TAG}_ba_iter_time.txt
cp $S/ba_batch_time.txt $P/${TAG}_ba_batch_time.txt
cp $S/ba_batch_11_windows_kernel_stats.csv $P/${TAG}_ba_batch_11_windows_kernel_stats.csv
cp $S/localba_wall.txt $P/${TAG}_localba_wall.txt
cp $S/lockstep_sweep.json $P/${TAG}_lockstep_sweep.json
cp $S/fuzz.log $P/${TAG}_fuzz.log
cp $S/pytest_gpu.log $P/${TAG}_pytest_gpu.log
ls $P | grep -v archive
