# K10 residual: two model accumulators at a time, three-step instantiation (4 waves per SIMD at R <= 12)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tools/gpu_steps.sh \
 "r04J_tests|600|python -m pytest tests/test_gpu_cp.py -m gpu -x -q -k 'fill_and_norm or rank_stream or residual'" \
 "r04J_k10|300|python tools/k10_probe.py 200 10 9"
