cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tools/gpu_steps.sh "r03q_n0|100|timeout -k 10 90 tools/place6_bench 5 0 0" "r03q_n1|100|timeout -k 10 90 tools/place6_bench 5 0 1" "r03q_n2|100|timeout -k 10 90 tools/place6_bench 5 0 2" "r03q_m0|100|timeout -k 10 90 tools/place6_bench 5 0 0" "r03q_m1|100|timeout -k 10 90 tools/place6_bench 5 0 1"
