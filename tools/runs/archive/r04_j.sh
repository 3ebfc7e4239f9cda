cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
CFG5="-model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32"
tools/gpu_steps.sh \
 "r04j_cfg5_nopipe|100|PPALS_TUCKER_PIPE=0 $B/test_ALS $CFG5 -filename gpurun_out/r04j_cfg5_nopipe.csv" \
 "r04j_cfg5_cu32|100|PPALS_SIDE_CUS=32 $B/test_ALS $CFG5 -filename gpurun_out/r04j_cfg5_cu32.csv" \
 "r04j_cfg5_cu64|100|PPALS_SIDE_CUS=64 $B/test_ALS $CFG5 -filename gpurun_out/r04j_cfg5_cu64.csv" \
 "r04j_cfg5_cu128|100|PPALS_SIDE_CUS=128 $B/test_ALS $CFG5 -filename gpurun_out/r04j_cfg5_cu128.csv" \
 "r04j_cfg5_nopipe2|100|PPALS_TUCKER_PIPE=0 $B/test_ALS $CFG5 -filename gpurun_out/r04j_cfg5_nopipe2.csv" \
 "r04j_trace_cu64|300|PPALS_SIDE_CUS=64 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04j_trace -o r04j -- $B/test_ALS $CFG5 -filename gpurun_out/r04j_cfg5_prof.csv"
