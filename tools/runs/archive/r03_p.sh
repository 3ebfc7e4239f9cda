cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="PPALS_DEBUG_ADDR=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-config-records"
tools/gpu_steps.sh "r03p_b1|200|$B" "r03p_b2|200|$B" "r03p_b3|200|$B" "r03p_b4|200|$B"
