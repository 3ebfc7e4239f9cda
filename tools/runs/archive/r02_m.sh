cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tools/gpu_steps.sh \
 "r02m_bench_tune|300|PPALS_DEBUG_ADDR=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline" \
 "r02m_bench_notune|300|PPALS_PLACE_TUNE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline" \
 "r02m_cfg4_tune|400|PPALS_DEBUG_ADDR=1 python bench.py --workload cp4_s400_r20 --steps 6 --warmup 2 --no-cpu-baseline" \
 "r02m_cfg4_notune|400|PPALS_PLACE_TUNE=0 python bench.py --workload cp4_s400_r20 --steps 6 --warmup 2 --no-cpu-baseline" \
 "r02m_tests|600|python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_cp.py tests/test_gpu_rccl.py -m gpu -x -q"
