# round 5, call b: the online placement choice on the GPU — CP parity tests, then the driver's bench command
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tools/gpu_steps.sh \
 "r05b_tests_cp|900|python -m pytest tests/test_gpu_cp.py tests/test_gpu_fullsize.py -m gpu -x -q -k 'not tucker and not order6' --durations=5" \
 "r05b_bench|600|python bench.py --gpus 1 --steps 20 --warmup 3"
tail -1 gpurun_out/r05b_bench.log > gpurun_out/r05b_bench.json
