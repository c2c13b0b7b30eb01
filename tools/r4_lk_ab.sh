#!/bin/bash
# A/B of LK kernel variants (build_var/*): tools/lk_micro.py at 2048 sequences (630 k keypoints), full pyramid and the 2-level pass
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
echo "== default"; python tools/lk_micro.py 2048 | grep -E "nbpyrlvl=3 max_iter=30|nbpyrlvl=3 max_iter= 0|nbpyrlvl=1 max_iter=30"
for v in "$@"; do
  echo "== $v"; OV2_LK_MICRO_LIB=$ROOT/build_var/$v/libov2slam_hip.so python tools/lk_micro.py 2048 | grep -E "nbpyrlvl=3 max_iter=30|nbpyrlvl=3 max_iter= 0|nbpyrlvl=1 max_iter=30"
done
