#!/bin/bash
# round-2 GPU session A: parity of the new tracker + single-sequence latency + kernel trace of the tracker loop
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r2a; mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_tracker.py tests/test_gpu_frontend.py tests/test_gpu_clahe.py tests/test_gpu_stereo.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 300 python tools/track_latency.py 400 > $OUT/latency.json 2> $OUT/latency.err; cat $OUT/latency.json; tail -3 $OUT/latency.err
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/tools/track_latency.py 100 > /dev/null 2> $OUT/trace.err
cd $ROOT
python - <<'PY'
import csv,glob,os
root=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for f in glob.glob(root+"/gpurun_out/r2a/trace/**/*_stats.csv",recursive=True):
    print(f)
    for r in list(csv.reader(open(f)))[:14]: print(r[:7])
PY
