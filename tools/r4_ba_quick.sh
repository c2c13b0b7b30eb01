#!/bin/bash
# quick BA check: the Cholesky A/B test + phase clocks + us per LM iteration
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
timeout 300 python -m pytest tests/test_gpu_ba.py -x -q --no-header -p no:cacheprovider -k "cholesky or config4 or protocol" 2>&1 | tail -4
bash tools/ba_ticks.sh 2>&1 | tail -4
python tools/ba_iter_time.py 2>&1 | tail -3
