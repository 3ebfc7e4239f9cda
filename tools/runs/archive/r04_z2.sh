cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
CFG5="-model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32"
tools/gpu_steps.sh \
 "r04z2_tucker_tests|900|python -m pytest tests/test_gpu_tucker.py tests/test_gpu_driver.py -m gpu -x -q" \
 "r04z2_cfg5|200|$B/test_ALS $CFG5 -filename gpurun_out/r04z2_cfg5_tucker.csv" \
 "r04z2_cfg5_log|200|PPALS_EIG_DEBUG=1 $B/test_ALS $CFG5 -filename gpurun_out/r04z2_cfg5_tucker_log.csv" \
 "r04z2_ppbench_tucker|300|$B/pp_bench -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -maxiter 5 -prec 32 -filename gpurun_out/r04z2_pp_bench_tucker.csv"
