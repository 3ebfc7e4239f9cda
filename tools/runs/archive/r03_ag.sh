cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tools/gpu_steps.sh "r03ag_tests|1000|python -m pytest tests -m gpu -x -q --durations=5" "r03ag_smoke|300|python -c 'import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")'"
