cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
tools/gpu_steps.sh \
 "r02h_tests|900|python -m pytest tests -m gpu -x -q" \
 "r02h_prof_pp1|300|rocprofv3 --kernel-trace --stats -d gpurun_out/r02h_prof_pp1 -o r02h -- $B/test_ALS -model CP -tensor r -dim 4 -size 200 -rank 10 -pp 1 -pp_res_tol 0.01 -maxiter 300 -prec 32 -filename gpurun_out/r02h_pp1.csv" \
 "r02h_tucker|300|$B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32 -filename gpurun_out/r02h_tucker40.csv"
