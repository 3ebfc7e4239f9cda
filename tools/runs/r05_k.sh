# round 5, call k: GPU suite + the fuzz campaign (1500 cases, skewed shapes included) before the evidence set
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tools/gpu_steps.sh \
 "r05k_tests|1100|python -m pytest tests -m gpu -x -q --durations=5" \
 "r05k_fuzz|1100|PPALS_FUZZ_CASES=1500 PPALS_FUZZ_SEED=6262626 python -m pytest tests/test_gpu_fuzz_campaign.py -x -q"
