# the GPU suite with the placement measurement forced onto every session (small tensors too) and every
# large-enough result stored non-temporally: the HIP side of the machinery on all test shapes
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tools/gpu_steps.sh \
 "r03ae_tests_forced_placement|1000|PPALS_PLACE_MIN_MB=0 PPALS_PLACE_SPACER_MB=64 PPALS_PLACE_PREFER_BLOCK=2 PPALS_SCAN_NT_MB=0 python -m pytest tests -m gpu -x -q"
