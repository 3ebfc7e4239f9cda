cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
tools/gpu_steps.sh \
 "r02f_tucker_dbg|300|PPALS_EIG_DEBUG=1 $B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 6 -prec 32 -filename gpurun_out/r02f_tucker.csv" \
 "r02f_tucker|300|$B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32 -filename gpurun_out/r02f_tucker40.csv" \
 "r02f_tests|600|python -m pytest tests/test_gpu_tucker.py tests/test_gpu_fullsize.py tests/test_gpu_rccl.py -m gpu -x -q"
