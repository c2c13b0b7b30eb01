#!/bin/bash
# BA iteration: GPU tests of the solver, then the kernel timeline of config 4 (mono) and the window, then us per LM iteration.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
timeout 600 python -m pytest tests/test_gpu_ba.py tests/test_gpu_xyz_ba.py tests/test_gpu_host_adapters.py -x -q --no-header -p no:cacheprovider 2>&1 | tail -15
bash tools/ba_timeline.sh config4_mono 2>&1 | sed -n 1,24p
bash tools/ba_timeline.sh window 2>&1 | sed -n 1,14p
python tools/ba_iter_time.py 2>&1 | tail -8
