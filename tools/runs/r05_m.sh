# round 5, call m: the persistent scan's grid on a P = 8 shard (3906 tiles): workgroups per CU 3 .. 40 (40 = one
# tile per workgroup at this size), same box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for m in 40 3 6 9 12 15 40 6; do
  echo "== PPALS_PERSIST_MULT=$m"
  PPALS_PERSIST_MULT=$m timeout -k 10 120 python tools/shard_probe.py 200 10 8 2>&1 | grep "P=8"
done > gpurun_out/r05m_persist_mult_shard.txt 2>&1
cat gpurun_out/r05m_persist_mult_shard.txt
