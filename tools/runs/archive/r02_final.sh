cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
RP="rocprofv3 --kernel-trace --stats --output-format csv"
tools/gpu_steps.sh \
 "r02G_tests|900|python -m pytest tests -m gpu -x -q" \
 "r02G_prof_bench|600|$RP -d gpurun_out/r02G_prof_bench -o r02G -- python3 bench.py --gpus 1 --steps 20 --warmup 5" \
 "r02G_pmc_fetch|400|rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r02G_pmc_fetch -o r02G -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline" \
 "r02G_pmc_write|400|rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r02G_pmc_write -o r02G -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline" \
 "r02G_prof_cfg4|600|$RP -d gpurun_out/r02G_prof_cfg4 -o r02G -- python3 bench.py --workload cp4_s400_r20 --steps 6 --warmup 2 --no-cpu-baseline" \
 "r02G_prof_cfg5|300|$RP -d gpurun_out/r02G_prof_cfg5 -o r02G -- $B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32 -filename gpurun_out/r02G_cfg5_tucker.csv" \
 "r02G_prof_pp1|300|$RP -d gpurun_out/r02G_prof_pp1 -o r02G -- $B/test_ALS -model CP -tensor r -dim 4 -size 200 -rank 10 -pp 1 -pp_res_tol 0.01 -maxiter 300 -prec 32 -filename gpurun_out/r02G_cfg3_pp1.csv" \
 "r02G_pp0|300|$B/test_ALS -model CP -tensor r -dim 4 -size 200 -rank 10 -pp 0 -maxiter 300 -prec 32 -filename gpurun_out/r02G_cfg3_pp0.csv" \
 "r02G_ppbench|200|$B/pp_bench -model CP -tensor r -dim 4 -size 200 -rank 10 -maxiter 5 -prec 32 -filename gpurun_out/r02G_pp_bench_cp.csv" \
 "r02G_ppbench_tucker|300|$B/pp_bench -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -maxiter 5 -prec 32 -filename gpurun_out/r02G_pp_bench_tucker.csv" \
 "r02G_cfg1|200|$B/test_ALS -model CP -tensor r -dim 3 -size 64 -rank 5 -pp 0 -filename gpurun_out/r02G_cfg1.csv" \
 "r02G_o1_make|400|python3 tools/make_o_file.py o1 /tmp/coil-100.bin 12 0.05" \
 "r02G_o1_cp|400|cd /tmp && $GRAFT_REPO_ROOT/$B/test_ALS -model CP -tensor o1 -dim 4 -rank 10 -pp 1 -maxiter 40 -prec 32 -filename $GRAFT_REPO_ROOT/gpurun_out/r02G_o1_cp_pp1.csv" \
 "r02G_o1_tucker|400|cd /tmp && $GRAFT_REPO_ROOT/$B/test_ALS -model Tucker -tensor o1 -dim 4 -pp 0 -maxiter 10 -prec 32 -filename $GRAFT_REPO_ROOT/gpurun_out/r02G_o1_tucker.csv; rm -f /tmp/coil-100.bin" \
 "r02G_o2_make|400|python3 tools/make_o_file.py o2 /tmp/time-lapse.bin 12 0.05" \
 "r02G_o2_cp|400|cd /tmp && $GRAFT_REPO_ROOT/$B/test_ALS -model CP -tensor o2 -dim 4 -rank 10 -pp 0 -maxiter 20 -prec 32 -filename $GRAFT_REPO_ROOT/gpurun_out/r02G_o2_cp_pp0.csv; rm -f /tmp/time-lapse.bin"
