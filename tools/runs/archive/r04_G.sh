cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tools/gpu_steps.sh \
 "r04G_tucker_tests|900|python -m pytest tests/test_gpu_tucker.py -m gpu -x -q -k 'between_32_and_64 or deferred'"
