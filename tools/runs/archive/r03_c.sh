cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
RP="rocprofv3 --kernel-trace --stats --output-format csv"
tools/gpu_steps.sh \
 "r03c_cp_tests|600|python -m pytest tests/test_gpu_cp.py tests/test_gpu_driver.py -m gpu -x -q" \
 "r03c_tucker_tests|600|python -m pytest tests/test_gpu_tucker.py tests/test_gpu_fullsize.py -m gpu -x -q -k 'tucker or eigen or hosvd or tall or chain'" \
 "r03c_update_bench|120|tools/update_bench" \
 "r03c_cfg5|200|$B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32 -filename gpurun_out/r03c_cfg5_tucker.csv" \
 "r03c_prof_cfg5|300|$RP -d gpurun_out/r03c_prof_cfg5 -o r03c -- $B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32 -filename gpurun_out/r03c_cfg5_tucker_prof.csv" \
 "r03c_ppbench|200|$B/pp_bench -model CP -tensor r -dim 4 -size 200 -rank 10 -maxiter 5 -prec 32 -filename gpurun_out/r03c_pp_bench_cp.csv" \
 "r03c_ppbench_nopresolve|200|PPALS_UPDATE_PRESOLVE=0 PPALS_UPDATE_MFMA=0 $B/pp_bench -model CP -tensor r -dim 4 -size 200 -rank 10 -maxiter 5 -prec 32 -filename gpurun_out/r03c_pp_bench_cp_old.csv" \
 "r03c_rccl_latency|120|python tools/rccl_latency.py"
