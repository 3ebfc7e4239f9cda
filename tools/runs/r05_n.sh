# round 5, call n: after the persistent-grid rule — GPU suite, smoke(), shard probe, the s = 100 cube, the driver's command
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tools/gpu_steps.sh \
 "r05n_tests|1100|python -m pytest tests -m gpu -x -q --durations=5" \
 "r05n_smoke|300|python -c 'import __graft_entry__ as g; g.smoke()'" \
 "r05n_shard_probe|300|python tools/shard_probe.py 200 10 8 && python tools/shard_probe.py 400 20 8 && python tools/shard_probe.py 100 10 1" \
 "r05n_shard_probe_old|300|PPALS_PERSIST_MULT=40 python tools/shard_probe.py 100 10 1" \
 "r05n_bench|600|python bench.py --gpus 1 --steps 20 --warmup 3"
tail -1 gpurun_out/r05n_bench.log > gpurun_out/r05n_bench.json
