#!/bin/bash
# A/B of the CLAHE histogram variants (build_var/<name> from tools/build_variant.sh): fused pre-processing per call vs input entropy
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r5_hist_ab; mkdir -p $OUT; export TMPDIR=/tmp; cd $ROOT
for v in default n1a0 n4a0 n1a1 n1a2 n4a1 n2a2 n8a2; do
  if [ $v = default ]; then unset OV2SLAM_HIP_LIB; else export OV2SLAM_HIP_LIB=$ROOT/build_var/$v/libov2slam_hip.so; fi
  echo "== $v" | tee -a $OUT/ab.txt
  timeout 200 python tools/pre_micro.py 4096 7 -1 euroc entropy 2>&1 | tail -1 | tee -a $OUT/ab.txt
done
