# round 5, call o: the side lane (prefetch of the next leaf's first contraction) — GPU suite with it on, then the A/B
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tools/gpu_steps.sh \
 "r05o_tests|1100|python -m pytest tests -m gpu -x -q --durations=5" \
 "r05o_lanes_ab|600|python tools/runs/lanes_ab.py 4"
