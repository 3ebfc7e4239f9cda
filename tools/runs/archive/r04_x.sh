# deferred tail with the Cholesky inside the multiplication + checks' G B on the second stream
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
RP="rocprofv3 --kernel-trace --stats --output-format csv"
CFG5="-model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32"
tools/gpu_steps.sh \
 "r04x_tucker_tests|900|python -m pytest tests/test_gpu_tucker.py tests/test_gpu_driver.py -m gpu -x -q" \
 "r04x_cfg5|200|$B/test_ALS $CFG5 -filename gpurun_out/r04x_cfg5_tucker.csv" \
 "r04x_cfg5_b|200|$B/test_ALS $CFG5 -filename gpurun_out/r04x_cfg5_tucker_b.csv" \
 "r04x_cfg5_log|200|PPALS_EIG_DEBUG=1 $B/test_ALS $CFG5 -filename gpurun_out/r04x_cfg5_tucker_log.csv" \
 "r04x_prof_cfg5|300|$RP -d gpurun_out/r04x_prof_cfg5 -o r04x -- $B/test_ALS $CFG5 -filename gpurun_out/r04x_cfg5_tucker_prof.csv"
