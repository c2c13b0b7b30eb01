#!/bin/bash
# A/B build of the library: one source file recompiled with extra flags, linked with the other objects of the regular build.
#   tools/build_variant.sh <name> <file.hip> [-DFLAG=VALUE ...]   ->  build_var/<name>/libov2slam_hip.so
# Select it at run time with OV2SLAM_HIP_LIB=<that path> (Python binding) -- tools/pre_micro.py, tools/lk_micro.py, bench.py.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); NAME=$1; SRC=$2; shift 2
make -C $ROOT/ov2slam_amd/csrc -j8 > /dev/null
OUT=$ROOT/build_var/$NAME; mkdir -p $OUT
BASE=$(basename $SRC .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function "$@" -c $ROOT/ov2slam_amd/csrc/$BASE.hip -o $OUT/$BASE.o
OBJS=$(ls $ROOT/ov2slam_amd/csrc/*.o | grep -v "/$BASE.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libov2slam_hip.so $OBJS $OUT/$BASE.o
echo $OUT/libov2slam_hip.so
