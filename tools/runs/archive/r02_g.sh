cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
tools/gpu_steps.sh \
 "r02g_tests|600|python -m pytest tests/test_gpu_tucker.py tests/test_gpu_cp.py tests/test_gpu_fullsize.py -m gpu -x -q" \
 "r02g_tucker|300|$B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32 -filename gpurun_out/r02g_tucker40.csv" \
 "r02g_prof_tucker|300|rocprofv3 --kernel-trace --stats -d gpurun_out/r02g_prof_tucker -o r02g -- $B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32 -filename gpurun_out/r02g_tucker40b.csv" \
 "r02g_prof_ppbench|300|rocprofv3 --kernel-trace --stats -d gpurun_out/r02g_prof_ppbench -o r02g -- $B/pp_bench -model CP -tensor r -dim 4 -size 200 -rank 10 -maxiter 5 -prec 32 -filename gpurun_out/r02g_pp_bench.csv"
