cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tools/gpu_steps.sh "r03y_r20|200|timeout -k 10 190 tools/scan2_bench 8000000 200 20 7" "r03y_r20b|200|timeout -k 10 190 tools/scan2_bench 8000000 200 20 7"
