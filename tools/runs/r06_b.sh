#!/bin/bash
# round 6, step b: wide-scan parity + the R = 100 sweep before/after + multirank logs
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06b; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_cp.py -x -q -m gpu -k "wide_scan" > $O/wide_tests.log 2>&1; echo "rc=$?" >> $O/wide_tests.log
tail -5 $O/wide_tests.log
PPALS_SCAN_WIDE=0 timeout -k 10 300 python tools/runs/r06_rank100.py 100 200 6 > $O/rank100_narrow.txt 2>&1; cat $O/rank100_narrow.txt
timeout -k 10 300 python tools/runs/r06_rank100.py 100 200 6 > $O/rank100_wide.txt 2>&1; cat $O/rank100_wide.txt
timeout -k 10 300 python tools/runs/r06_rank100.py 128 200 6 > $O/rank128_wide.txt 2>&1; cat $O/rank128_wide.txt
