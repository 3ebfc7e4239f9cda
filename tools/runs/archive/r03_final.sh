# the evidence set of round 3 (one gpurun call): GPU tests, the driver's bench command under
# rocprofv3, cfg5 / PP / class-API runs. usage: tools/runs/r03_final.sh   (from the repository root)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
RP="rocprofv3 --kernel-trace --stats --output-format csv"
tools/gpu_steps.sh \
 "r03H_tests|900|python -m pytest tests -m gpu -x -q --durations=8" \
 "r03H_bench|500|python bench.py --gpus 1 --steps 20 --warmup 5" \
 "r03H_prof_bench|600|$RP -d gpurun_out/r03H_prof_bench -o r03H -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-config-records" \
 "r03H_prof_cfg5|300|$RP -d gpurun_out/r03H_prof_cfg5 -o r03H -- $B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32 -filename gpurun_out/r03H_cfg5_tucker_prof.csv" \
 "r03H_cfg5|200|$B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32 -filename gpurun_out/r03H_cfg5_tucker.csv" \
 "r03H_cfg5_round2_tail|200|PPALS_EIG_FUSED=0 $B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32 -filename gpurun_out/r03H_cfg5_tucker_round2_eigenstep.csv" \
 "r03H_ppbench|200|$B/pp_bench -model CP -tensor r -dim 4 -size 200 -rank 10 -maxiter 5 -prec 32 -filename gpurun_out/r03H_pp_bench_cp.csv" \
 "r03H_ppbench_tucker|300|$B/pp_bench -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -maxiter 5 -prec 32 -filename gpurun_out/r03H_pp_bench_tucker.csv" \
 "r03H_shard_probe|300|python tools/shard_probe.py 200 10 1,2,4,8 msdt" \
 "r03H_shard_probe_cfg4|400|python tools/shard_probe.py 400 20 8 msdt"
