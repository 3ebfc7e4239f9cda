cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tools/gpu_steps.sh "r03n_p1|150|tools/place4_bench 5 0" "r03n_p2|150|tools/place4_bench 5 1" "r03n_p3|150|tools/place4_bench 5 0" "r03n_p4|150|tools/place4_bench 5 1"
