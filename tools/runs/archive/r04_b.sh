# round 4, call b: deferred acceptance of eigen-steps — GPU tests of the Tucker path, cfg5 A/B, trace
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
CFG5="-model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32"
tools/gpu_steps.sh \
 "r04b_tests_tucker|600|python -m pytest tests/test_gpu_tucker.py -x -q --durations=5" \
 "r04b_cfg5_defer|100|$B/test_ALS $CFG5 -filename gpurun_out/r04b_cfg5_defer.csv && $B/test_ALS $CFG5 -filename gpurun_out/r04b_cfg5_defer2.csv" \
 "r04b_cfg5_nodefer|100|PPALS_EIG_DEFER=0 $B/test_ALS $CFG5 -filename gpurun_out/r04b_cfg5_nodefer.csv && PPALS_EIG_DEFER=0 $B/test_ALS $CFG5 -filename gpurun_out/r04b_cfg5_nodefer2.csv" \
 "r04b_cfg5_log|100|PPALS_EIG_DEBUG=1 $B/test_ALS $CFG5 -filename gpurun_out/r04b_cfg5_log.csv" \
 "r04b_trace_cfg5|300|rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04b_trace_cfg5 -o r04b -- $B/test_ALS $CFG5 -filename gpurun_out/r04b_cfg5_prof.csv" \
 "r04b_tests_full|600|python -m pytest tests/test_gpu_fullsize.py -x -q -k cfg5"
