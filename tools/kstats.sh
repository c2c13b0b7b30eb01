#!/bin/bash
# Per-kernel durations of a short bench run (rocprofv3 --kernel-trace --stats), top rows printed; run through gpurun.
# With OV2SLAM_HIP_LIB=<variant .so> it times an A/B or knock-out build of the library.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/ks; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python $ROOT/bench.py --steps 40 --warmup 10 --no-extras --no-cpu-baseline > $OUT/b.json 2> $OUT/err
python - <<'PY'
import csv,glob,os
f=glob.glob(os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/ks/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.reader(open(f)))[1:8]: print(r[0][:50], r[1], r[3], r[5], r[6])
PY
