cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
RP="rocprofv3 --kernel-trace --stats --output-format csv"
tools/gpu_steps.sh \
 "r03f_tucker_tests|600|python -m pytest tests/test_gpu_tucker.py tests/test_gpu_fullsize.py -m gpu -x -q -k 'tucker or eigen or hosvd or tall or chain'" \
 "r03f_cp_tests|600|python -m pytest tests/test_gpu_cp.py tests/test_gpu_driver.py -m gpu -x -q" \
 "r03f_cfg5|200|$B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32 -filename gpurun_out/r03f_cfg5_tucker.csv" \
 "r03f_prof_cfg5|300|$RP -d gpurun_out/r03f_prof_cfg5 -o r03f -- $B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32 -filename gpurun_out/r03f_cfg5_tucker_prof.csv" \
 "r03f_script6|300|$B/test_ALS -model Tucker -tensor r2 -dim 6 -size 50 -rank 6 -pp 0 -maxiter 10 -prec 32 -filename gpurun_out/r03f_tucker_order6.csv" \
 "r03f_lr_dt|300|$B/run -model CP -tensor r -dim 4 -size 200 -rank 10 -pp 2 -updaterank 2 -maxiter 20 -resprint 5 -prec 32 -filename gpurun_out/r03f_run_pp2.csv" \
 "r03f_lr_msdt|300|$B/run -model CP -tensor r -dim 4 -size 200 -rank 10 -pp 3 -updaterank 2 -maxiter 20 -resprint 5 -prec 32 -filename gpurun_out/r03f_run_pp3.csv" \
 "r03f_run_msdt|300|$B/run -model CP -tensor r -dim 4 -size 200 -rank 10 -pp 1 -maxiter 20 -resprint 5 -prec 32 -filename gpurun_out/r03f_run_pp1.csv"
