# round 4, call c: lazy core + checks on the second stream
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
CFG5="-model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32"
tools/gpu_steps.sh \
 "r04c_tests_tucker|600|python -m pytest tests/test_gpu_tucker.py -x -q --durations=5" \
 "r04c_cfg5|100|$B/test_ALS $CFG5 -filename gpurun_out/r04c_cfg5.csv && $B/test_ALS $CFG5 -filename gpurun_out/r04c_cfg5_2.csv" \
 "r04c_trace_cfg5|300|rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04c_trace_cfg5 -o r04c -- $B/test_ALS $CFG5 -filename gpurun_out/r04c_cfg5_prof.csv" \
 "r04c_tests_full|600|python -m pytest tests/test_gpu_fullsize.py -x -q -k cfg5"
