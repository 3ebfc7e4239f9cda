cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tools/gpu_steps.sh "r03l_p1|120|tools/place2_bench" "r03l_p2|120|tools/place2_bench" "r03l_p3|120|tools/place2_bench" "r03l_p4|120|tools/place2_bench" "r03l_p5|120|tools/place2_bench"
