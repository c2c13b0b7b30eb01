#!/bin/bash
# knock-out sweep of the strip kernel: tools/build_variant.sh koN clahe.hip -DCS_KO=N beforehand; times pre-processing without the LUT kernel
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
for v in "$@"; do echo -n "$v: "; OV2SLAM_HIP_LIB=$ROOT/build_var/$v/libov2slam_hip.so python tools/pre_micro.py 4096 6 -1 2 2>&1 | tail -1; done
