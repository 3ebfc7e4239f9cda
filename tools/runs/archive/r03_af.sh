cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
tools/gpu_steps.sh \
 "r03af_lr_tests|600|python -m pytest tests/test_gpu_cp.py tests/test_gpu_driver.py -q -x -k 'low_rank or class_api'" \
 "r03af_run_pp3_rnd|120|$B/run -model CP -tensor r -dim 4 -size 200 -rank 10 -pp 3 -updaterank 2 -randomsvd 1 -maxiter 20 -prec 32 -filename gpurun_out/r03af_run_pp3_rnd.csv" \
 "r03af_run_pp3|120|$B/run -model CP -tensor r -dim 4 -size 200 -rank 10 -pp 3 -updaterank 2 -randomsvd 0 -maxiter 20 -prec 32 -filename gpurun_out/r03af_run_pp3.csv"
