#!/bin/bash
# Tucker on the reference scripts' order-6 shape: chain order of the first-level products
B=pairwise-perturbation_amd/bin
T6="-model Tucker -tensor r2 -dim 6 -size 50 -rank 6 -pp 0 -maxiter 14 -prec 32"
exec tools/gpu_steps.sh \
 "r02tk_tests|600|PPALS_TUCKER_CHAIN=desc timeout -k 10 500 python -m pytest tests/test_gpu_tucker.py -m gpu -x -q" \
 "r02tk_auto|500|timeout -k 10 450 $B/test_ALS $T6 -filename gpurun_out/r02tk_tucker6_auto.csv" \
 "r02tk_asc|500|PPALS_TUCKER_CHAIN=asc timeout -k 10 450 $B/test_ALS $T6 -filename gpurun_out/r02tk_tucker6_asc.csv"
