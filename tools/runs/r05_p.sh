# round 5, call p: a second fuzz campaign on the last commit (another seed, 3000 cases)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tools/gpu_steps.sh \
 "r05p_fuzz|1150|PPALS_FUZZ_CASES=3000 PPALS_FUZZ_SEED=7373737 python -m pytest tests/test_gpu_fuzz_campaign.py -x -q"
