#!/bin/bash
# round 2, step r: the thin route of the Tucker factor update on the coil-100 shape
B=pairwise-perturbation_amd/bin
RP="rocprofv3 --kernel-trace --stats --output-format csv"
exec tools/gpu_steps.sh \
 "r02r_tests|900|python -m pytest tests/test_gpu_tucker.py -m gpu -x -q" \
 "r02r_o1_make|400|python3 tools/make_o_file.py o1 /tmp/coil-100.bin 12 0.05" \
 "r02r_o1_tucker_thin|400|cd /tmp && $GRAFT_REPO_ROOT/$B/test_ALS -model Tucker -tensor o1 -dim 4 -pp 0 -maxiter 10 -prec 32 -filename $GRAFT_REPO_ROOT/gpurun_out/r02r_o1_tucker_thin.csv" \
 "r02r_o1_tucker_big|400|cd /tmp && PPALS_TUCKER_THIN=0 $GRAFT_REPO_ROOT/$B/test_ALS -model Tucker -tensor o1 -dim 4 -pp 0 -maxiter 10 -prec 32 -filename $GRAFT_REPO_ROOT/gpurun_out/r02r_o1_tucker_big.csv" \
 "r02r_o1_tucker_pp|400|cd /tmp && $GRAFT_REPO_ROOT/$B/test_ALS -model Tucker -tensor o1 -dim 4 -pp 1 -maxiter 30 -prec 32 -filename $GRAFT_REPO_ROOT/gpurun_out/r02r_o1_tucker_pp.csv" \
 "r02r_prof_o1|400|cd /tmp && $RP -d $GRAFT_REPO_ROOT/gpurun_out/r02r_prof_o1 -o r02r -- $GRAFT_REPO_ROOT/$B/test_ALS -model Tucker -tensor o1 -dim 4 -pp 0 -maxiter 10 -prec 32 -filename $GRAFT_REPO_ROOT/gpurun_out/r02r_o1_tucker_prof.csv; rm -f /tmp/coil-100.bin"
