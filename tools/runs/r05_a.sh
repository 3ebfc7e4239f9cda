# round 5, call a: CU-mask feasibility (tools/cumask_bench) + kernel-trace timelines of the P = 1 sweep and
# of the P = 8 shard's rank-local sweep (what the latency chain is made of, launch by launch)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
RP="rocprofv3 --kernel-trace --output-format csv"
tools/gpu_steps.sh \
 "r05a_cumask|300|tools/cumask_bench 6.4" \
 "r05a_trace_p1|300|$RP -d gpurun_out/r05a_trace_p1 -o t -- python3 tools/shard_probe.py 200 10 1" \
 "r05a_trace_p8|300|$RP -d gpurun_out/r05a_trace_p8 -o t -- python3 tools/shard_probe.py 200 10 8"
for d in r05a_trace_p1 r05a_trace_p8; do
  f=$(find gpurun_out/$d -name '*kernel_trace.csv' | head -1)
  python3 tools/trace_timeline.py "$f" k_scan_suffix 20 90 > gpurun_out/${d}_timeline.txt 2>&1
  rm -rf gpurun_out/$d
done
