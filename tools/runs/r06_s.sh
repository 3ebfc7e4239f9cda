#!/bin/bash
# round 6, step s: timeline of a time-lapse HOOI sweep (core ranks 10, 100, 100, 5): where the stream idles
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=r06s
tools/gpu_steps.sh \
 "${T}_trace_tl|400|rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${T}_trace -o t -- python3 tools/runs/real_tucker_probe.py timelapse 6"
f=$(find gpurun_out/${T}_trace -name '*kernel_trace.csv' | head -1)
python3 tools/trace_timeline.py "$f" k_jacobi_onesided 10 330 > gpurun_out/${T}_tl_sweep_timeline.txt 2>&1
rm -rf gpurun_out/${T}_trace
tail -45 gpurun_out/${T}_tl_sweep_timeline.txt
