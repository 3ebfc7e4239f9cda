# round 5, call e: GPU suite after the long-mode / PP-chain / fused-Normalize changes, the real-data extents
# again (CP line + kernel stats of the exact sweeps + Tucker kernel stats), shard probe, the driver's command
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
RP="rocprofv3 --kernel-trace --stats --output-format csv"
tools/gpu_steps.sh \
 "r05e_tests|1100|python -m pytest tests -m gpu -x -q --durations=8" \
 "r05e_coil100|400|PPALS_PLACE_MIN_MB=1000 python bench.py --workload coil100 --steps 20 --warmup 3" \
 "r05e_timelapse|400|PPALS_PLACE_MIN_MB=1000 python bench.py --workload timelapse --steps 20 --warmup 3" \
 "r05e_prof_coil|300|$RP -d gpurun_out/r05e_prof_coil -o p -- python3 bench.py --workload coil100 --steps 30 --warmup 3 --no-config-records" \
 "r05e_prof_tl|300|$RP -d gpurun_out/r05e_prof_tl -o p -- python3 bench.py --workload timelapse --steps 30 --warmup 3 --no-config-records" \
 "r05e_prof_tk_coil|300|$RP -d gpurun_out/r05e_prof_tk_coil -o p -- python3 tools/runs/real_tucker_probe.py coil100" \
 "r05e_prof_tk_tl|300|$RP -d gpurun_out/r05e_prof_tk_tl -o p -- python3 tools/runs/real_tucker_probe.py timelapse" \
 "r05e_shard_probe|300|python tools/shard_probe.py 200 10 8 && python tools/shard_probe.py 400 20 8" \
 "r05e_bench|600|python bench.py --gpus 1 --steps 20 --warmup 3"
for n in coil100 timelapse bench; do tail -1 gpurun_out/r05e_$n.log > gpurun_out/r05e_$n.json; done
for d in prof_coil prof_tl prof_tk_coil prof_tk_tl; do
  f=$(find gpurun_out/r05e_$d -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r05e_${d}_kernel_stats.csv; rm -rf gpurun_out/r05e_$d
done
