cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
RP="rocprofv3 --kernel-trace --stats --output-format csv"
tools/gpu_steps.sh \
 "r03d_cp_tests|600|python -m pytest tests/test_gpu_cp.py tests/test_gpu_driver.py -m gpu -x -q" \
 "r03d_tucker_tests|600|python -m pytest tests/test_gpu_tucker.py tests/test_gpu_fullsize.py -m gpu -x -q -k 'tucker or eigen or hosvd or tall or chain or cfg3 or pp_driver'" \
 "r03d_cfg5|200|$B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32 -filename gpurun_out/r03d_cfg5_tucker.csv" \
 "r03d_prof_cfg5|300|$RP -d gpurun_out/r03d_prof_cfg5 -o r03d -- $B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32 -filename gpurun_out/r03d_cfg5_tucker_prof.csv" \
 "r03d_ppbench|200|$B/pp_bench -model CP -tensor r -dim 4 -size 200 -rank 10 -maxiter 5 -prec 32 -filename gpurun_out/r03d_pp_bench_cp.csv" \
 "r03d_ppbench_nograph|200|PPALS_GRAPH=0 $B/pp_bench -model CP -tensor r -dim 4 -size 200 -rank 10 -maxiter 5 -prec 32 -filename gpurun_out/r03d_pp_bench_cp_nograph.csv" \
 "r03d_pp1|200|$B/test_ALS -model CP -tensor r -dim 4 -size 200 -rank 10 -pp 1 -pp_res_tol 0.01 -maxiter 300 -prec 32 -filename gpurun_out/r03d_cfg3_pp1.csv" \
 "r03d_pp1_nograph|200|PPALS_GRAPH=0 $B/test_ALS -model CP -tensor r -dim 4 -size 200 -rank 10 -pp 1 -pp_res_tol 0.01 -maxiter 300 -prec 32 -filename gpurun_out/r03d_cfg3_pp1_nograph.csv" \
 "r03d_prof_pp1|300|$RP -d gpurun_out/r03d_prof_pp1 -o r03d -- $B/test_ALS -model CP -tensor r -dim 4 -size 200 -rank 10 -pp 1 -pp_res_tol 0.01 -maxiter 300 -prec 32 -filename gpurun_out/r03d_cfg3_pp1_prof.csv"
