#!/bin/bash
# round 2, step y: eigen-step with one read-back (cfg5)
B=pairwise-perturbation_amd/bin
RP="rocprofv3 --kernel-trace --stats --output-format csv"
T5="-model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32"
exec tools/gpu_steps.sh \
 "r02y_tests|900|python -m pytest tests/test_gpu_tucker.py tests/test_gpu_fuzz_campaign.py -m gpu -x -q" \
 "r02y_cfg5|300|PPALS_EIG_DEBUG=1 $B/test_ALS $T5 -filename gpurun_out/r02y_cfg5_tucker.csv" \
 "r02y_cfg5_nodebug|300|$B/test_ALS $T5 -filename gpurun_out/r02y_cfg5_tucker_noprof.csv" \
 "r02y_prof_cfg5|300|$RP -d gpurun_out/r02y_prof_cfg5 -o r02y -- $B/test_ALS $T5 -filename gpurun_out/r02y_cfg5_tucker_prof.csv" \
 "r02y_ppbench_tucker|300|$B/pp_bench -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -maxiter 5 -prec 32 -filename gpurun_out/r02y_pp_bench_tucker.csv"
