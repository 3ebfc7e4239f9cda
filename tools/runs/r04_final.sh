# the evidence set of round 4 (one gpurun call): GPU tests, the driver's bench command plain and under
# rocprofv3 --kernel-trace --stats, cfg5 stats + step log, pp_bench CP / Tucker, rank-100 step log.
# usage: tools/runs/r04_final.sh   (from the repository root)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
RP="rocprofv3 --kernel-trace --stats --output-format csv"
CFG5="-model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32"
tools/gpu_steps.sh \
 "r04Z_tests|1100|python -m pytest tests -m gpu -x -q --durations=8" \
 "r04Z_bench|600|python bench.py --gpus 1 --steps 20 --warmup 5" \
 "r04Z_prof_bench|600|$RP -d gpurun_out/r04Z_prof_bench -o r04Z -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-config-records --no-pmc" \
 "r04Z_prof_cfg5|300|$RP -d gpurun_out/r04Z_prof_cfg5 -o r04Z -- $B/test_ALS $CFG5 -filename gpurun_out/r04Z_cfg5_tucker_prof.csv" \
 "r04Z_cfg5|200|$B/test_ALS $CFG5 -filename gpurun_out/r04Z_cfg5_tucker.csv" \
 "r04Z_cfg5_nodefer|200|PPALS_EIG_DEFER=0 $B/test_ALS $CFG5 -filename gpurun_out/r04Z_cfg5_tucker_nodefer.csv" \
 "r04Z_cfg5_log|200|PPALS_EIG_DEBUG=1 $B/test_ALS $CFG5 -filename gpurun_out/r04Z_cfg5_tucker_log.csv" \
 "r04Z_ppbench|200|$B/pp_bench -model CP -tensor r -dim 4 -size 200 -rank 10 -maxiter 5 -prec 32 -filename gpurun_out/r04Z_pp_bench_cp.csv" \
 "r04Z_ppbench_tucker|300|$B/pp_bench -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -maxiter 5 -prec 32 -filename gpurun_out/r04Z_pp_bench_tucker.csv" \
 "r04Z_rank100|300|python tools/runs/big_rank_probe.py 1"
