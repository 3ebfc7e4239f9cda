cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
CFG5="-model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32"
tools/gpu_steps.sh \
 "r04A_tucker_tests|900|python -m pytest tests/test_gpu_tucker.py tests/test_gpu_driver.py -m gpu -x -q" \
 "r04A_cfg5|200|$B/test_ALS $CFG5 -filename gpurun_out/r04A_cfg5_tucker.csv" \
 "r04A_cfg5_log|200|PPALS_EIG_DEBUG=1 $B/test_ALS $CFG5 -filename gpurun_out/r04A_cfg5_tucker_log.csv" \
 "r04A_cfg5_b|200|$B/test_ALS $CFG5 -filename gpurun_out/r04A_cfg5_tucker_b.csv" \
 "r04A_prof_cfg5|300|rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04A_prof_cfg5 -o r04A -- $B/test_ALS $CFG5 -filename gpurun_out/r04A_cfg5_tucker_prof.csv"
