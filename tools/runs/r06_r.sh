#!/bin/bash
# round 6, step r: timeline of an R = 100 sweep after the host synchronisation left the mode update
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=r06r
tools/gpu_steps.sh \
 "${T}_trace_r100|300|rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${T}_trace -o t -- python3 tools/runs/r06_sweep_times.py 100 200 4"
f=$(find gpurun_out/${T}_trace -name '*kernel_trace.csv' | head -1)
python3 tools/trace_timeline.py "$f" k_scan_wide 14 48 > gpurun_out/${T}_r100_timeline.txt 2>&1
rm -rf gpurun_out/${T}_trace
cat gpurun_out/${T}_r100_timeline.txt
