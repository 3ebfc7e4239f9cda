cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tools/gpu_steps.sh "r03m_p1|150|tools/place3_bench" "r03m_p2|150|tools/place3_bench" "r03m_p3|150|tools/place3_bench"
