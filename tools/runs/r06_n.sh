#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=r06n
RP="rocprofv3 --kernel-trace --stats --output-format csv"
tools/gpu_steps.sh \
 "${T}_prof_tl|600|$RP -d gpurun_out/${T}_prof_tl -o p -- python3 tools/runs/real_tucker_probe.py timelapse 20"
f=$(find gpurun_out/${T}_prof_tl -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/${T}_tk_tl_kernel_stats.csv
rm -rf gpurun_out/${T}_prof_tl
head -22 gpurun_out/${T}_tk_tl_kernel_stats.csv | cut -c1-100,100-400 | awk -F'",' '{print $1}' | cut -c1-90 > /dev/null
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r06n_tk_tl_kernel_stats.csv")))
for r in rows[:22]:
    print(f'{r["Name"].split("(")[0][-60:]:60s} calls {r["Calls"]:>6} total_ms {int(r["TotalDurationNs"])/1e6:9.2f} avg_us {float(r["AverageNs"])/1e3:9.1f} {r["Percentage"]}%')
PY
