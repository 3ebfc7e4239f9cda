cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tools/gpu_steps.sh \
 "r03j_b1|200|PPALS_DEBUG_ADDR=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-config-records" \
 "r03j_b2|200|PPALS_DEBUG_ADDR=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-config-records" \
 "r03j_b3|200|PPALS_DEBUG_ADDR=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-config-records" \
 "r03j_b4|200|PPALS_DEBUG_ADDR=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-config-records" \
 "r03j_b5|200|PPALS_DEBUG_ADDR=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-config-records"
