cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
tools/gpu_steps.sh \
 "r02o_tests|600|python -m pytest tests/test_gpu_tucker.py tests/test_gpu_driver.py tests/test_gpu_fullsize.py -m gpu -x -q" \
 "r02o_prof_tucker|300|rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02o_prof_cfg5 -o r02o -- $B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32 -filename gpurun_out/r02o_cfg5_tucker.csv" \
 "r02o_tucker|300|$B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32 -filename gpurun_out/r02o_cfg5_tucker_noprof.csv"
