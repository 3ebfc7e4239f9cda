# round 4, call d: the symmetric product — latency kernel, LDS-tiled kernel, persistent chain with grid barriers
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tools/gpu_steps.sh \
 "r04d_nsprod_400|120|tools/nsprod_bench 400" \
 "r04d_nsprod_640|120|tools/nsprod_bench 640" \
 "r04d_nsprod_896|120|tools/nsprod_bench 896" \
 "r04d_nsprod_1344|120|tools/nsprod_bench 1344"
