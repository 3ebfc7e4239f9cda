cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tools/gpu_steps.sh "r03ad_mfma64|60|timeout -k 5 50 tools/mfma64_rate"
