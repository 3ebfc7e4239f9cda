#!/bin/bash
# Tucker on the reference scripts' order-6 shape
B=pairwise-perturbation_amd/bin
T6="-model Tucker -tensor r2 -dim 6 -size 50 -rank 6 -pp 0 -maxiter 14 -prec 32"
exec tools/gpu_steps.sh \
 "r02tk_tests|600|timeout -k 10 500 python -m pytest tests/test_gpu_tucker.py tests/test_gpu_fuzz_campaign.py tests/test_gpu_driver.py -m gpu -x -q" \
 "r02tk_auto|500|timeout -k 10 450 $B/test_ALS $T6 -filename gpurun_out/r02tk_tucker6_auto.csv" \
 "r02tk_cold|500|PPALS_EIG_FAST=0 timeout -k 10 450 $B/test_ALS $T6 -filename gpurun_out/r02tk_tucker6_coldeig.csv" \
 "r02tk_pp|500|timeout -k 10 450 $B/test_ALS -model Tucker -tensor r2 -dim 6 -size 50 -rank 6 -pp 1 -maxiter 30 -prec 32 -filename gpurun_out/r02tk_tucker6_pp.csv"
