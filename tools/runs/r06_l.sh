#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=r06l
tools/gpu_steps.sh \
 "${T}_wide_tests|600|PPALS_FUZZ_CASES=24 python -m pytest tests/test_gpu_cp.py -x -q -m gpu -k wide_scan" \
 "${T}_r100_k1|300|PPALS_MSDT_ROOTS=1 python3 tools/runs/r06_rank100.py 100 200 6" \
 "${T}_r100|300|python3 tools/runs/r06_rank100.py 100 200 6" \
 "${T}_r128_k1|300|PPALS_MSDT_ROOTS=1 python3 tools/runs/r06_rank100.py 128 200 6" \
 "${T}_r70_k1|300|PPALS_MSDT_ROOTS=1 python3 tools/runs/r06_rank100.py 70 200 6"
