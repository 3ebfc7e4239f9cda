# round 5, call c: whole GPU suite with the measured-bar log, the real-data extents, the one-rank rehearsal of
# the N > 1 bench line, and the driver's command
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -f gpurun_out/r05c_fp32_bars.jsonl
tools/gpu_steps.sh \
 "r05c_tests|1100|PPALS_BAR_LOG=$GRAFT_REPO_ROOT/gpurun_out/r05c_fp32_bars.jsonl python -m pytest tests -m gpu -x -q --durations=8" \
 "r05c_coil100|400|PPALS_PLACE_MIN_MB=1000 python bench.py --workload coil100 --steps 20 --warmup 3" \
 "r05c_timelapse|400|PPALS_PLACE_MIN_MB=1000 python bench.py --workload timelapse --steps 20 --warmup 3" \
 "r05c_bench_forcecomm|900|PPALS_FORCE_COMM=1 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline" \
 "r05c_bench|600|python bench.py --gpus 1 --steps 20 --warmup 3"
for n in coil100 timelapse bench_forcecomm bench; do tail -1 gpurun_out/r05c_$n.log > gpurun_out/r05c_$n.json; done
