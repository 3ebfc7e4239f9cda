cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O="--no-cpu-baseline --no-config-records"
tools/gpu_steps.sh "r03x_s400r10|300|python bench.py --workload cp4_s400_r10 --steps 4 --warmup 1 $O" "r03x_s200r20|200|python bench.py --workload cp4_s200_r20 --steps 20 --warmup 5 $O" "r03x_s400r20|300|python bench.py --workload cp4_s400_r20 --steps 4 --warmup 1 $O" "r03x_s200r10|200|python bench.py --steps 20 --warmup 5 $O"
