cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
CFG5="-model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32"
tools/gpu_steps.sh \
 "r04q_trace_cfg5|300|rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04q_trace_cfg5 -o r04q -- $B/test_ALS $CFG5 -filename gpurun_out/r04q_cfg5_prof.csv" \
 "r04q_tucker|400|python -m pytest tests/test_gpu_tucker.py tests/test_gpu_fullsize.py -x -q -k 'tucker or Tucker or cfg5 or rank or flat or eigen or projector or hosvd'"
