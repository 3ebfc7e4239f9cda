#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=r06h
tools/gpu_steps.sh \
 "${T}_tests|600|python -m pytest tests/test_gpu_cp.py -x -q -m gpu -k 'wide_scan or gram_system or above_64'" \
 "${T}_r100|300|python3 tools/runs/r06_rank100.py 100 200 6" \
 "${T}_r100_scalar|300|PPALS_GJ_SCALAR=1 python3 tools/runs/r06_rank100.py 100 200 6" \
 "${T}_trace_r100|300|rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${T}_trace -o t -- python3 tools/runs/r06_sweep_times.py 100 200 4"
f=$(find gpurun_out/${T}_trace -name '*kernel_trace.csv' | head -1)
python3 tools/trace_timeline.py "$f" k_scan_wide 14 45 > gpurun_out/${T}_r100_timeline.txt 2>&1
rm -rf gpurun_out/${T}_trace
tail -22 gpurun_out/${T}_r100_timeline.txt
