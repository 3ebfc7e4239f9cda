cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
RP="rocprofv3 --kernel-trace --stats --output-format csv"
tools/gpu_steps.sh \
 "r03b_tucker_tests|600|python -m pytest tests/test_gpu_tucker.py tests/test_gpu_fullsize.py -m gpu -x -q -k 'tucker or eigen or hosvd or tall or chain'" \
 "r03b_cfg5_dbg|200|PPALS_EIG_DEBUG=1 $B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32 -filename gpurun_out/r03b_cfg5_tucker.csv" \
 "r03b_cfg5_old|200|PPALS_EIG_FUSED=0 $B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32 -filename gpurun_out/r03b_cfg5_tucker_unfused.csv" \
 "r03b_prof_cfg5|300|$RP -d gpurun_out/r03b_prof_cfg5 -o r03b -- $B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32 -filename gpurun_out/r03b_cfg5_tucker_prof.csv" \
 "r03b_bench|500|python bench.py --gpus 1 --steps 20 --warmup 5"
