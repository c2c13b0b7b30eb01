#!/bin/bash
# round-5 BA A/B (run through gpurun): fused kernel tails on / off -- parity tests, then the iteration times
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r5_ba_ab; mkdir -p $OUT; export TMPDIR=/tmp
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_ba.py tests/test_gpu_xyz_ba.py tests/test_gpu_host_adapters.py tests/test_reference_factors.py -m gpu -x -q 2>&1 | tail -5 | tee $OUT/tests.txt
for f in 0 1 0 1; do
  echo "OV2_BA_FUSE=$f" | tee -a $OUT/iter.txt
  OV2_BA_FUSE=$f timeout 300 python tools/ba_iter_time.py 2>&1 | tail -3 | tee -a $OUT/iter.txt
done
