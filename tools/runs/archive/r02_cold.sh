#!/bin/bash
# Ritz-guided cold start of the Tucker eigen-step: tests, cfg5, the coil-100 shape
B=pairwise-perturbation_amd/bin
T5="-model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32"
exec tools/gpu_steps.sh \
 "r02c_tests|800|PPALS_FUZZ_CASES=300 timeout -k 10 700 python -m pytest tests/test_gpu_tucker.py tests/test_gpu_fuzz_campaign.py tests/test_gpu_driver.py tests/test_gpu_fullsize.py -m gpu -x -q -k 'tucker or Tucker or hosvd'" \
 "r02c_cfg5|300|timeout -k 10 200 $B/test_ALS $T5 -filename gpurun_out/r02c_cfg5.csv" \
 "r02c_cfg5_cold0|300|PPALS_EIG_COLD=0 timeout -k 10 200 $B/test_ALS $T5 -filename gpurun_out/r02c_cfg5_cold0.csv" \
 "r02c_o1_make|400|python3 tools/make_o_file.py o1 /tmp/coil-100.bin 12 0.05" \
 "r02c_o1|400|cd /tmp && timeout -k 10 300 $GRAFT_REPO_ROOT/$B/test_ALS -model Tucker -tensor o1 -dim 4 -pp 0 -maxiter 10 -prec 32 -filename $GRAFT_REPO_ROOT/gpurun_out/r02c_o1_tucker.csv" \
 "r02c_o1_cold0|400|cd /tmp && PPALS_EIG_COLD=0 timeout -k 10 300 $GRAFT_REPO_ROOT/$B/test_ALS -model Tucker -tensor o1 -dim 4 -pp 0 -maxiter 10 -prec 32 -filename $GRAFT_REPO_ROOT/gpurun_out/r02c_o1_tucker_cold0.csv; rm -f /tmp/coil-100.bin"
