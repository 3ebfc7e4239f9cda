cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
CFG5="-model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32"
tools/gpu_steps.sh \
 "r04i_cfg5_shared|100|$B/test_ALS $CFG5 -filename gpurun_out/r04i_cfg5.csv && $B/test_ALS $CFG5 -filename gpurun_out/r04i_cfg5_2.csv" \
 "r04i_cfg5_own|100|PPALS_SIDE_OWN_STREAM=1 $B/test_ALS $CFG5 -filename gpurun_out/r04i_cfg5_own.csv" \
 "r04i_cfg5_nopipe|100|PPALS_TUCKER_PIPE=0 $B/test_ALS $CFG5 -filename gpurun_out/r04i_cfg5_nopipe.csv" \
 "r04i_trace_cfg5|300|rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04i_trace_cfg5 -o r04i -- $B/test_ALS $CFG5 -filename gpurun_out/r04i_cfg5_prof.csv"
