# the evidence set of round 5 (one gpurun call): GPU tests, the driver's bench command plain and under
# rocprofv3 --kernel-trace --stats, cfg5 stats + step log, pp_bench CP / Tucker, the P = 8 shard's launch
# timeline, the real-data extents.   usage: tools/runs/r05_final.sh   (from the repository root)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
RP="rocprofv3 --kernel-trace --stats --output-format csv"
CFG5="-model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32"
T=r05Z
tools/gpu_steps.sh \
 "${T}_tests|1100|python -m pytest tests -m gpu -x -q --durations=8" \
 "${T}_bench|600|python bench.py --gpus 1 --steps 20 --warmup 3" \
 "${T}_prof_bench|600|$RP -d gpurun_out/${T}_prof_bench -o p -- python3 bench.py --gpus 1 --steps 20 --warmup 3 --no-config-records --no-pmc" \
 "${T}_prof_cfg5|300|$RP -d gpurun_out/${T}_prof_cfg5 -o p -- $B/test_ALS $CFG5 -filename gpurun_out/${T}_cfg5_tucker_prof.csv" \
 "${T}_cfg5|200|$B/test_ALS $CFG5 -filename gpurun_out/${T}_cfg5_tucker.csv" \
 "${T}_cfg5_log|200|PPALS_EIG_DEBUG=1 $B/test_ALS $CFG5 -filename gpurun_out/${T}_cfg5_tucker_log.csv" \
 "${T}_ppbench|200|$B/pp_bench -model CP -tensor r -dim 4 -size 200 -rank 10 -maxiter 5 -prec 32 -filename gpurun_out/${T}_pp_bench_cp.csv" \
 "${T}_ppbench_tucker|300|$B/pp_bench -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -maxiter 5 -prec 32 -filename gpurun_out/${T}_pp_bench_tucker.csv" \
 "${T}_trace_p8|300|rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${T}_trace_p8 -o t -- python3 tools/shard_probe.py 200 10 8" \
 "${T}_shard_probe|300|python tools/shard_probe.py 200 10 1,8 && python tools/shard_probe.py 400 20 8"
for n in bench; do tail -1 gpurun_out/${T}_$n.log > gpurun_out/${T}_$n.json; done
grep -a -o '{"metric.*' gpurun_out/${T}_prof_bench.log | tail -1 > gpurun_out/${T}_bench_under_rocprof.json
f=$(find gpurun_out/${T}_prof_bench -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/${T}_bench_kernel_stats.csv
f=$(find gpurun_out/${T}_prof_bench -name '*kernel_trace.csv' | head -1); python3 tools/trace_timed_launches.py gpurun_out/${T}_bench_under_rocprof.json "$f" > gpurun_out/${T}_timed_launches.txt 2>&1
f=$(find gpurun_out/${T}_prof_cfg5 -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/${T}_cfg5_kernel_stats.csv
f=$(find gpurun_out/${T}_trace_p8 -name '*kernel_trace.csv' | head -1); python3 tools/trace_timeline.py "$f" k_scan_suffix 20 90 > gpurun_out/${T}_trace_p8_timeline.txt 2>&1
grep "ppals eig" gpurun_out/${T}_cfg5_log.log | tail -40 > gpurun_out/${T}_cfg5_step_log.txt
rm -rf gpurun_out/${T}_prof_bench gpurun_out/${T}_prof_cfg5 gpurun_out/${T}_trace_p8
