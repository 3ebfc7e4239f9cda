#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=r06m
tools/gpu_steps.sh \
 "${T}_tucker_tests|900|python -m pytest tests/test_gpu_tucker.py -x -q -m gpu" \
 "${T}_timelapse|600|python bench.py --workload timelapse" \
 "${T}_coil100|600|python bench.py --workload coil100"
tail -1 gpurun_out/${T}_coil100.log > gpurun_out/${T}_coil100.json
tail -1 gpurun_out/${T}_timelapse.log > gpurun_out/${T}_timelapse.json
python3 - <<'PY'
import json
for n in ("coil100","timelapse"):
    d=json.load(open(f"gpurun_out/r06m_{n}.json"))
    t=d.get("tucker", d.get("sub_records",{}).get("tucker"))
    print(n, "sweeps/s", round(d["value"],1), "hosvd_ms", round(t["hosvd_ms"],1), "ms_per_hooi_sweep", round(t["ms_per_hooi_sweep"],3))
PY
