cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tools/gpu_steps.sh \
 "r02d_bench|420|python bench.py --steps 20 --warmup 5" \
 "r02d_probe_cfg2|300|python tools/shard_probe.py 200 10 1,2,4,8" \
 "r02d_probe_cfg4|600|python tools/shard_probe.py 400 20 1,2,4,8" \
 "r02d_probe_cfg4_dt|600|python tools/shard_probe.py 400 20 1,8 dt"
