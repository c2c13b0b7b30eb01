#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r2d; mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_xyz_ba.py tests/test_gpu_ba.py tests/test_gpu_detect.py tests/test_gpu_tracker.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -25 $OUT/pytest.log
