cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
tools/gpu_steps.sh \
 "r03ak_tests|1000|python -m pytest tests -m gpu -x -q" \
 "r03ak_ppbench|200|$B/pp_bench -model CP -tensor r -dim 4 -size 200 -rank 10 -maxiter 7 -prec 32 -filename gpurun_out/r03ak_pp_bench_cp.csv" \
 "r03ak_pp1|200|rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03ak_prof -o r03ak -- $B/test_ALS -model CP -tensor r -dim 4 -size 200 -rank 10 -pp 1 -pp_res_tol 0.01 -maxiter 300 -prec 32 -filename gpurun_out/r03ak_pp1.csv"
