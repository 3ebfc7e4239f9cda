# round 5, call d: whole GPU suite with the measured-bar log, the placement A/B, the real-data extents
# after the short-mode roots were excluded, where -pp 1 spends its time there, the P = 8 shard probe
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -f gpurun_out/r05d_fp32_bars.jsonl
RP="rocprofv3 --kernel-trace --stats --output-format csv"
tools/gpu_steps.sh \
 "r05d_tests|1100|PPALS_BAR_LOG=$GRAFT_REPO_ROOT/gpurun_out/r05d_fp32_bars.jsonl python -m pytest tests -m gpu -x -q --durations=8" \
 "r05d_place_ab|300|python tools/runs/place_ab.py 8 250" \
 "r05d_coil100|400|PPALS_PLACE_MIN_MB=1000 python bench.py --workload coil100 --steps 20 --warmup 3" \
 "r05d_timelapse|400|PPALS_PLACE_MIN_MB=1000 python bench.py --workload timelapse --steps 20 --warmup 3" \
 "r05d_pp_probe|400|$RP -d gpurun_out/r05d_pp_probe -o p -- python3 tools/runs/real_pp_probe.py coil100" \
 "r05d_shard_probe|300|python tools/shard_probe.py 200 10 8 && python tools/shard_probe.py 400 20 8"
for n in coil100 timelapse; do tail -1 gpurun_out/r05d_$n.log > gpurun_out/r05d_$n.json; done
f=$(find gpurun_out/r05d_pp_probe -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r05d_pp_probe_kernel_stats.csv; rm -rf gpurun_out/r05d_pp_probe
