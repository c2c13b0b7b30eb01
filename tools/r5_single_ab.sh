#!/bin/bash
# single-camera latency with one / four histogram copies in k_clahe_lut (build_var/n1 = -DCH_NCOPY=1)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
for v in default n1 default n1; do
  if [ $v = default ]; then unset OV2SLAM_HIP_LIB; else export OV2SLAM_HIP_LIB=$ROOT/build_var/$v/libov2slam_hip.so; fi
  echo -n "== $v: "; python tools/track_latency.py 600 2>&1 | python -c "
import sys, json
t = sys.stdin.read(); d = json.loads(t[t.index('{'):])
print({k: round(v['median_ms'], 4) for k, v in d.items()})"
done
