#!/bin/bash
# round 6, step k: the LDS-staged scan at <= 64 columns (experiment): cfg4 and cfg2 two-node tree scans
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=r06k
( for W in 0 17; do echo "== cfg4 (s=400 R=20) PPALS_WIDE_MIN_COLS=$W"; PPALS_WIDE_MIN_COLS=$W timeout -k 10 280 python3 tools/runs/r06_rank100.py 20 400 4 2>&1 | grep "R="; done
  for W in 0 1; do echo "== cfg2 (s=200 R=10) PPALS_WIDE_MIN_COLS=$W"; PPALS_WIDE_MIN_COLS=$W timeout -k 10 200 python3 tools/runs/r06_rank100.py 10 200 6 2>&1 | grep "R="; done
  for W in 0 33; do echo "== s=200 R=40 PPALS_WIDE_MIN_COLS=$W"; PPALS_WIDE_MIN_COLS=$W timeout -k 10 200 python3 tools/runs/r06_rank100.py 40 200 6 2>&1 | grep "R="; done
  for W in 0 49; do echo "== s=200 R=64 PPALS_WIDE_MIN_COLS=$W"; PPALS_WIDE_MIN_COLS=$W timeout -k 10 200 python3 tools/runs/r06_rank100.py 64 200 6 2>&1 | grep "R="; done ) > gpurun_out/${T}_wide_low.txt 2>&1
cat gpurun_out/${T}_wide_low.txt
