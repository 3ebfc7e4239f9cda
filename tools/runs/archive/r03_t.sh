cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tools/gpu_steps.sh \
 "r03t_tests|1000|python -m pytest tests -m gpu -x -q --durations=8" \
 "r03t_bench|500|python bench.py --gpus 1 --steps 20 --warmup 5"
