cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
C="$B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32 -filename"
tools/gpu_steps.sh \
 "r03ai_tucker_tests|600|python -m pytest tests/test_gpu_tucker.py tests/test_gpu_fullsize.py -x -q" \
 "r03ai_c0|100|$C gpurun_out/r03ai_c0.csv" "r03ai_c1|100|$C gpurun_out/r03ai_c1.csv" "r03ai_c2|100|$C gpurun_out/r03ai_c2.csv"
