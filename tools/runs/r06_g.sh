#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=r06g
tools/gpu_steps.sh \
 "${T}_wide_tests|600|python -m pytest tests/test_gpu_cp.py -x -q -m gpu -k wide_scan" \
 "${T}_r100|300|python3 tools/runs/r06_rank100.py 100 200 6" \
 "${T}_r128|300|python3 tools/runs/r06_rank100.py 128 200 6" \
 "${T}_r70|300|python3 tools/runs/r06_rank100.py 70 200 6" \
 "${T}_r128_x2|300|PPALS_WIDE_EXP=2 python3 tools/runs/r06_rank100.py 128 200 4" \
 "${T}_r128_x14|300|PPALS_WIDE_EXP=14 python3 tools/runs/r06_rank100.py 128 200 4"
