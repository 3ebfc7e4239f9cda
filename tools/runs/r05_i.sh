# round 5, call i: after the NaN-safe ranking of the Jacobi kernels — GPU suite (incl. the tail-mode scan and the
# exactly-low-rank Tucker tests), the time-lapse line, Tucker on both real-data extents, the shard-rows probe,
# cfg5 through the driver's command
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
RP="rocprofv3 --kernel-trace --stats --output-format csv"
tools/gpu_steps.sh \
 "r05i_tests|1100|python -m pytest tests -m gpu -x -q --durations=8" \
 "r05i_timelapse|400|PPALS_PLACE_MIN_MB=1000 python bench.py --workload timelapse --steps 20 --warmup 3" \
 "r05i_prof_tk_coil|300|$RP -d gpurun_out/r05i_prof_tk_coil -o p -- python3 tools/runs/real_tucker_probe.py coil100" \
 "r05i_prof_tk_tl|300|$RP -d gpurun_out/r05i_prof_tk_tl -o p -- python3 tools/runs/real_tucker_probe.py timelapse" \
 "r05i_shard_rows|300|python tools/runs/shard_rows_probe.py" \
 "r05i_bench|600|python bench.py --gpus 1 --steps 20 --warmup 3"
for n in timelapse bench; do tail -1 gpurun_out/r05i_$n.log > gpurun_out/r05i_$n.json; done
for d in prof_tk_coil prof_tk_tl; do
  f=$(find gpurun_out/r05i_$d -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r05i_${d}_kernel_stats.csv; rm -rf gpurun_out/r05i_$d
done
