cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
RP="rocprofv3 --kernel-trace --stats --output-format csv"
tools/gpu_steps.sh \
 "r03g_tucker_tests|600|python -m pytest tests/test_gpu_tucker.py tests/test_gpu_fullsize.py tests/test_gpu_driver.py -m gpu -x -q -k 'tucker or Tucker or eigen or hosvd or tall or chain or every_tensor or pp_bench'" \
 "r03g_cfg5|200|$B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32 -filename gpurun_out/r03g_cfg5_tucker.csv" \
 "r03g_cfg5_nolazy|200|PPALS_EIG_LAZY=0 $B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32 -filename gpurun_out/r03g_cfg5_tucker_nolazy.csv" \
 "r03g_prof_cfg5|300|$RP -d gpurun_out/r03g_prof_cfg5 -o r03g -- $B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32 -filename gpurun_out/r03g_cfg5_tucker_prof.csv" \
 "r03g_cfg5_f64|200|$B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 64 -filename gpurun_out/r03g_cfg5_tucker_f64.csv" \
 "r03g_cfg5_pp|200|$B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 1 -maxiter 40 -prec 32 -filename gpurun_out/r03g_cfg5_tucker_pp1.csv"
