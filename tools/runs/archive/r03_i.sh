cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
tools/gpu_steps.sh \
 "r03i_tests|900|python -m pytest tests -m gpu -x -q" \
 "r03i_bench|500|python bench.py --gpus 1 --steps 20 --warmup 5" \
 "r03i_bench_nopresolve|300|PPALS_UPDATE_PRESOLVE=0 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-config-records" \
 "r03i_bench2|300|python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-config-records" \
 "r03i_shard_probe|300|python tools/shard_probe.py 200 10 8 msdt" \
 "r03i_shard_probe_nopresolve|300|PPALS_UPDATE_PRESOLVE=0 python tools/shard_probe.py 200 10 8 msdt" \
 "r03i_shard_probe_cfg4|400|python tools/shard_probe.py 400 20 8 msdt"
