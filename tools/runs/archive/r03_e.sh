cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
RP="rocprofv3 --kernel-trace --stats --output-format csv"
tools/gpu_steps.sh \
 "r03e_cp_tests|600|python -m pytest tests/test_gpu_cp.py tests/test_gpu_driver.py tests/test_gpu_padded.py -m gpu -x -q" \
 "r03e_tucker_tests|600|python -m pytest tests/test_gpu_tucker.py tests/test_gpu_fullsize.py -m gpu -x -q -k 'tucker or eigen or hosvd or tall or chain or cfg3 or pp_driver'" \
 "r03e_nsprod|100|tools/nsprod_bench 400" \
 "r03e_update_bench|120|tools/update_bench" \
 "r03e_cfg5|200|$B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32 -filename gpurun_out/r03e_cfg5_tucker.csv" \
 "r03e_ppbench|200|$B/pp_bench -model CP -tensor r -dim 4 -size 200 -rank 10 -maxiter 5 -prec 32 -filename gpurun_out/r03e_pp_bench_cp.csv" \
 "r03e_ppbench_nofuse|200|PPALS_UPDATE_FUSE_NORM=0 $B/pp_bench -model CP -tensor r -dim 4 -size 200 -rank 10 -maxiter 5 -prec 32 -filename gpurun_out/r03e_pp_bench_cp_nofuse.csv" \
 "r03e_pmc_fetch_cfg4|500|rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r03e_pmc_fetch_cfg4 -o r03e -- python3 bench.py --workload cp4_s400_r20 --steps 3 --warmup 1 --no-cpu-baseline --no-config-records" \
 "r03e_pmc_write_cfg4|500|rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r03e_pmc_write_cfg4 -o r03e -- python3 bench.py --workload cp4_s400_r20 --steps 3 --warmup 1 --no-cpu-baseline --no-config-records" \
 "r03e_pmc_fetch|400|rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r03e_pmc_fetch -o r03e -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config-records" \
 "r03e_pmc_write|400|rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r03e_pmc_write -o r03e -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config-records"
