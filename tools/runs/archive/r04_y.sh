# merged elimination + multiplication (k_rmult_chol, 1024 threads) and the wait-value hand-over probe
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
RP="rocprofv3 --kernel-trace --stats --output-format csv"
CFG5="-model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32"
tools/gpu_steps.sh \
 "r04y_tucker_tests|900|python -m pytest tests/test_gpu_tucker.py tests/test_gpu_driver.py -m gpu -x -q" \
 "r04y_cfg5|200|$B/test_ALS $CFG5 -filename gpurun_out/r04y_cfg5_tucker.csv" \
 "r04y_cfg5_b|200|$B/test_ALS $CFG5 -filename gpurun_out/r04y_cfg5_tucker_b.csv" \
 "r04y_prof_cfg5|300|$RP -d gpurun_out/r04y_prof_cfg5 -o r04y -- $B/test_ALS $CFG5 -filename gpurun_out/r04y_cfg5_tucker_prof.csv" \
 "r04y_waitvalue|90|hipcc --offload-arch=gfx950 -O3 -o /tmp/waitvalue_bench tools/waitvalue_bench.hip && timeout -k 5 45 /tmp/waitvalue_bench"
