#!/bin/bash
# round-5 probes (run through gpurun): pre-processing vs histogram entropy, localBA wall-clock laps, BA iteration times
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r5_probe; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
timeout 300 python tools/pre_micro.py 4096 6 -1 euroc entropy 2>&1 | tail -2 | tee $OUT/pre_entropy.txt
timeout 200 python tools/pre_micro.py 4096 6 0 euroc entropy 2>&1 | tail -1 | tee -a $OUT/pre_entropy.txt
OV2_DEBUG=1 timeout 300 python tools/localba_wall.py 2>&1 | tail -40 | tee $OUT/localba_wall.txt
timeout 300 python tools/ba_iter_time.py 2>&1 | tail -12 | tee $OUT/ba_iter_time.txt
