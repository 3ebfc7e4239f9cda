# round 5, call r: the last commit once more — GPU suite, smoke(), the driver's command
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tools/gpu_steps.sh \
 "r05r_tests|1100|python -m pytest tests -m gpu -x -q --durations=5" \
 "r05r_smoke|300|python -c 'import __graft_entry__ as g; g.smoke()'" \
 "r05r_bench|600|python bench.py --gpus 1 --steps 20 --warmup 3"
tail -1 gpurun_out/r05r_bench.log > gpurun_out/r05r_bench.json
