# power steps counted from the measured movement of the dominant eigenvector
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
CFG5="-model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32"
tools/gpu_steps.sh \
 "r04P_tucker_tests|900|python -m pytest tests/test_gpu_tucker.py tests/test_gpu_driver.py tests/test_gpu_fullsize.py -m gpu -x -q -k 'tucker or Tucker or cfg5'" \
 "r04P_cfg5|200|$B/test_ALS $CFG5 -filename gpurun_out/r04P_cfg5_tucker.csv" \
 "r04P_cfg5_log|200|PPALS_EIG_DEBUG=1 $B/test_ALS $CFG5 -filename gpurun_out/r04P_cfg5_tucker_log.csv" \
 "r04P_cfg5_b|200|$B/test_ALS $CFG5 -filename gpurun_out/r04P_cfg5_tucker_b.csv" \
 "r04P_prof_cfg5|300|rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04P_prof_cfg5 -o r04P -- $B/test_ALS $CFG5 -filename gpurun_out/r04P_cfg5_tucker_prof.csv"
