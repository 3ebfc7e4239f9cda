# k_rmult_chol with the series of S^-1/2 in four fused stages, workgroup size by the number of columns
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
CFG5="-model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32"
tools/gpu_steps.sh \
 "r04B_rcbench|120|hipcc --offload-arch=gfx950 -O3 -o /tmp/rcb tools/rmult_chol_bench.hip && /tmp/rcb 400 21 0.05 && /tmp/rcb 400 21 0.05 1024 && /tmp/rcb 400 21 0.05 512 && /tmp/rcb 1344 21 0.05 && /tmp/rcb 400 40 0.05 && /tmp/rcb 400 56 0.05" \
 "r04B_tucker_tests|900|python -m pytest tests/test_gpu_tucker.py tests/test_gpu_driver.py -m gpu -x -q" \
 "r04B_cfg5|200|$B/test_ALS $CFG5 -filename gpurun_out/r04B_cfg5_tucker.csv" \
 "r04B_cfg5_log|200|PPALS_EIG_DEBUG=1 $B/test_ALS $CFG5 -filename gpurun_out/r04B_cfg5_tucker_log.csv" \
 "r04B_cfg5_b|200|$B/test_ALS $CFG5 -filename gpurun_out/r04B_cfg5_tucker_b.csv" \
 "r04B_prof_cfg5|300|rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04B_prof_cfg5 -o r04B -- $B/test_ALS $CFG5 -filename gpurun_out/r04B_cfg5_tucker_prof.csv"
