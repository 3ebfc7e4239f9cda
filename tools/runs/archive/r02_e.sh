cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
tools/gpu_steps.sh \
 "r02e_tests|900|python -m pytest tests -m gpu -x -q" \
 "r02e_tucker|300|$B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32 -filename gpurun_out/r02e_tucker.csv" \
 "r02e_tucker_slow|300|PPALS_EIG_FAST=0 $B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32 -filename gpurun_out/r02e_tucker_slow.csv" \
 "r02e_ppbench|200|$B/pp_bench -model CP -tensor r -dim 4 -size 200 -rank 10 -maxiter 5 -prec 32 -filename gpurun_out/r02e_pp_bench.csv" \
 "r02e_pp1|300|$B/test_ALS -model CP -tensor r -dim 4 -size 200 -rank 10 -pp 1 -pp_res_tol 0.01 -maxiter 300 -prec 32 -filename gpurun_out/r02e_pp1.csv" \
 "r02e_pp0|300|$B/test_ALS -model CP -tensor r -dim 4 -size 200 -rank 10 -pp 0 -maxiter 300 -prec 32 -filename gpurun_out/r02e_pp0.csv"
