# round 5, call l: the timed launches of the headline inside a kernel trace of the driver's command (the step of
# r05_final.sh whose post-processing failed), on the final commit
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
RP="rocprofv3 --kernel-trace --stats --output-format csv"
T=r05Z2
tools/gpu_steps.sh \
 "${T}_prof_bench|600|$RP -d gpurun_out/${T}_prof_bench -o p -- python3 bench.py --gpus 1 --steps 20 --warmup 3 --no-config-records --no-pmc" \
 "${T}_bench|600|python bench.py --gpus 1 --steps 20 --warmup 3"
tail -1 gpurun_out/${T}_bench.log > gpurun_out/${T}_bench.json
grep -a -o '{"metric.*' gpurun_out/${T}_prof_bench.log | tail -1 > gpurun_out/${T}_bench_under_rocprof.json
f=$(find gpurun_out/${T}_prof_bench -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/${T}_bench_kernel_stats.csv
f=$(find gpurun_out/${T}_prof_bench -name '*kernel_trace.csv' | head -1); python3 tools/trace_timed_launches.py gpurun_out/${T}_bench_under_rocprof.json "$f" > gpurun_out/${T}_timed_launches.txt 2>&1
rm -rf gpurun_out/${T}_prof_bench
cat gpurun_out/${T}_timed_launches.txt
