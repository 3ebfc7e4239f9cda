# the evidence set of round 5, second call: the real-data extents, the placement A/B, the fuzz campaign.
# usage: tools/runs/r05_final2.sh   (from the repository root)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
RP="rocprofv3 --kernel-trace --stats --output-format csv"
CFG5="-model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32"
T=r05Z
tools/gpu_steps.sh \
 "${T}_coil100|400|PPALS_PLACE_MIN_MB=1000 python bench.py --workload coil100 --steps 20 --warmup 3" \
 "${T}_timelapse|400|PPALS_PLACE_MIN_MB=1000 python bench.py --workload timelapse --steps 20 --warmup 3" \
 "${T}_place_ab|300|python tools/runs/place_ab.py 8 250" \
 "${T}_fuzz|1100|PPALS_FUZZ_CASES=1500 PPALS_FUZZ_SEED=6262626 python -m pytest tests/test_gpu_fuzz_campaign.py -x -q"
for n in coil100 timelapse; do tail -1 gpurun_out/${T}_$n.log > gpurun_out/${T}_$n.json; done
