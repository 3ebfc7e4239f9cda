cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
tools/gpu_steps.sh \
 "r02i_tests|900|python -m pytest tests/test_gpu_cp.py tests/test_gpu_fullsize.py -m gpu -x -q" \
 "r02i_prof_pp0|300|rocprofv3 --kernel-trace --stats -d gpurun_out/r02i_prof_pp0 -o r02i -- $B/test_ALS -model CP -tensor r -dim 4 -size 200 -rank 10 -pp 0 -maxiter 60 -prec 32 -filename gpurun_out/r02i_pp0.csv" \
 "r02i_prof_pp0_r20|300|rocprofv3 --kernel-trace --stats -d gpurun_out/r02i_prof_pp0_r20 -o r02i -- $B/test_ALS -model CP -tensor r -dim 4 -size 200 -rank 20 -pp 0 -maxiter 30 -prec 32 -filename gpurun_out/r02i_pp0_r20.csv"
