cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tools/gpu_steps.sh "r03q_x1|100|timeout -k 10 90 tools/place6_bench 7 0 3" "r03q_x2|100|timeout -k 10 90 tools/place6_bench 7 0 3"
