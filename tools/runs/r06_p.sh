#!/bin/bash
# round 6, step p: the CLI at the reference's default rank (s/2): -pp 0 against -pp 1, s = 200, R = 100
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
T=r06p
COMMON="-model CP -tensor r -dim 4 -size 200 -rank 100 -maxiter 60 -resprint 10 -prec 32 -tol 1e-6"
tools/gpu_steps.sh \
 "${T}_pp0|300|$B/test_ALS $COMMON -pp 0 -filename gpurun_out/${T}_pp0.csv" \
 "${T}_pp1|300|$B/test_ALS $COMMON -pp 1 -pp_res_tol 0.05 -filename gpurun_out/${T}_pp1.csv"
echo "--- pp0"; cat gpurun_out/${T}_pp0.csv; echo "--- pp1"; cat gpurun_out/${T}_pp1.csv
