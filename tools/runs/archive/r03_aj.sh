cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tools/gpu_steps.sh "r03aj_fuzz|900|PPALS_FUZZ_CASES=1500 PPALS_FUZZ_SEED=31337 python -m pytest tests/test_gpu_fuzz_campaign.py -q -x"
