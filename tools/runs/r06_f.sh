#!/bin/bash
# round 6, step f: what the parts of the wide scan's loop cost (bench-only variants, WRONG results)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=r06f
for R in 100 128; do for X in 0 2 4 8 14; do
  echo "== R=$R exp=$X"; PPALS_WIDE_EXP=$X timeout -k 10 200 python3 tools/runs/r06_rank100.py $R 200 4 2>&1 | grep "dt:" ; done; done > gpurun_out/${T}_exp.txt 2>&1
cat gpurun_out/${T}_exp.txt
