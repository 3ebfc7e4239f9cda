#!/bin/bash
# the evidence set of round 6 (one gpurun call): GPU tests, the driver's bench command plain and under
# rocprofv3 --kernel-trace --stats, cfg4 and the R = 100 sweep under rocprofv3, cfg5 stats, the wide-regime
# fuzz campaign.   usage: tools/runs/r06_final.sh [tag]   (from the repository root, inside gpurun)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
RP="rocprofv3 --kernel-trace --stats --output-format csv"
CFG5="-model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32"
T=${1:-r06Z}
tools/gpu_steps.sh \
 "${T}_tests|1100|PPALS_MULTIRANK_LOG=gpurun_out/${T}_multirank python -m pytest tests -m gpu -x -q --durations=8" \
 "${T}_bench|600|python bench.py --gpus 1 --steps 20 --warmup 3" \
 "${T}_prof_bench|600|$RP -d gpurun_out/${T}_prof_bench -o p -- python3 bench.py --gpus 1 --steps 20 --warmup 3 --no-config-records --no-pmc" \
 "${T}_prof_cfg4|400|$RP -d gpurun_out/${T}_prof_cfg4 -o p -- python3 tools/runs/r06_rank100.py 20 400 4" \
 "${T}_prof_r100|300|$RP -d gpurun_out/${T}_prof_r100 -o p -- python3 tools/runs/r06_rank100.py 100 200 6" \
 "${T}_prof_cfg5|300|$RP -d gpurun_out/${T}_prof_cfg5 -o p -- $B/test_ALS $CFG5 -filename gpurun_out/${T}_cfg5_tucker_prof.csv" \
 "${T}_cfg5|200|$B/test_ALS $CFG5 -filename gpurun_out/${T}_cfg5_tucker.csv" \
 "${T}_fuzz_wide|900|PPALS_FUZZ_CASES=40 PPALS_FUZZ_SEED=6060606 python -m pytest tests/test_gpu_cp.py -x -q -m gpu -k wide_scan_random"
tail -1 gpurun_out/${T}_bench.log > gpurun_out/${T}_bench.json
grep -a -o '{"metric.*' gpurun_out/${T}_prof_bench.log | tail -1 > gpurun_out/${T}_bench_under_rocprof.json
for n in bench cfg4 r100 cfg5; do
  f=$(find gpurun_out/${T}_prof_$n -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/${T}_${n}_kernel_stats.csv
done
f=$(find gpurun_out/${T}_prof_bench -name '*kernel_trace.csv' | head -1); python3 tools/trace_timed_launches.py gpurun_out/${T}_bench_under_rocprof.json "$f" > gpurun_out/${T}_timed_launches.txt 2>&1
rm -rf gpurun_out/${T}_prof_bench gpurun_out/${T}_prof_cfg4 gpurun_out/${T}_prof_r100 gpurun_out/${T}_prof_cfg5
