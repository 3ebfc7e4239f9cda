#!/bin/bash
# round 6, step e: wide scan v2 (register ring 3 deep, occupancy-aware k-split): tests + R = 100 / 128 / 70 sweeps
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=r06e
tools/gpu_steps.sh \
 "${T}_wide_tests|600|python -m pytest tests/test_gpu_cp.py -x -q -m gpu -k wide_scan" \
 "${T}_r100|300|python3 tools/runs/r06_rank100.py 100 200 6" \
 "${T}_r128|300|python3 tools/runs/r06_rank100.py 128 200 6" \
 "${T}_r70|300|python3 tools/runs/r06_rank100.py 70 200 6" \
 "${T}_r100_narrow|300|PPALS_SCAN_WIDE=0 python3 tools/runs/r06_rank100.py 100 200 6"
