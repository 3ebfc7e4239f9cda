# round 4, call h: prefetched first contractions on the side lane (Tucker order 3)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
CFG5="-model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32"
tools/gpu_steps.sh \
 "r04h_tests_tucker|600|python -m pytest tests/test_gpu_tucker.py -x -q --durations=5" \
 "r04h_cfg5|100|$B/test_ALS $CFG5 -filename gpurun_out/r04h_cfg5.csv && $B/test_ALS $CFG5 -filename gpurun_out/r04h_cfg5_2.csv" \
 "r04h_cfg5_nopipe|100|PPALS_TUCKER_PIPE=0 $B/test_ALS $CFG5 -filename gpurun_out/r04h_cfg5_nopipe.csv" \
 "r04h_trace_cfg5|300|rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04h_trace_cfg5 -o r04h -- $B/test_ALS $CFG5 -filename gpurun_out/r04h_cfg5_prof.csv" \
 "r04h_tests_full|600|python -m pytest tests/test_gpu_fullsize.py -x -q -k 'cfg5 or order6'"
