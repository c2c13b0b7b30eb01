cd $GRAFT_REPO_ROOT
P=$1; shift
for k in "$@"; do
  if [ $k = 0 ]; then unset OV2SLAM_HIP_LIB; else export OV2SLAM_HIP_LIB=$GRAFT_REPO_ROOT/build_var/${P}_ko$k/libov2slam_hip.so; fi
  echo "== KO $k"; bash tools/kstat_cmd.sh ko$k python $GRAFT_REPO_ROOT/tools/detect_batch_time.py 4096 1 2>&1 | grep "k_mineig\|k_corner\|k_grid"
done
