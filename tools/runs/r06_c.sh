#!/bin/bash
# round 6, step c: where an R = 100 sweep goes (kernel stats), wide-scan tests
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
RP="rocprofv3 --kernel-trace --stats --output-format csv"
T=r06c
tools/gpu_steps.sh \
 "${T}_wide_tests|600|python -m pytest tests/test_gpu_cp.py -x -q -m gpu -k wide_scan" \
 "${T}_prof_r100|300|$RP -d gpurun_out/${T}_prof_r100 -o p -- python3 tools/runs/r06_rank100.py 100 200 6"
f=$(find gpurun_out/${T}_prof_r100 -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/${T}_r100_kernel_stats.csv
rm -rf gpurun_out/${T}_prof_r100
cut -c1-150 gpurun_out/${T}_r100_kernel_stats.csv | head -30
