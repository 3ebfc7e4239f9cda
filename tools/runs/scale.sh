#!/bin/bash
# The strong-scaling curve of BASELINE configs[1] (cfg2: CP order-4 s=200 R=10) and configs[3]
# (cfg4: s=400 R=20) on ONE 8-GPU node: exactly the command lines the round driver uses for
# `bench.py --gpus N` (one process per GPU, RCCL over xGMI), for N = 1, 2, 4, 8. Not runnable on
# the one-GPU boxes of this pool; kept so that the curve can be reproduced wherever 8 GPUs are.
#   usage: tools/runs/scale.sh [outdir]            (from the repository root)
# Every line of <outdir>/scale_*.jsonl is one bench.py JSON record: `value` = sweeps/s of the WHOLE
# job (strong scaling: same tensor, leading-mode shards), `rccl_ranks` = the communicator size the
# engine reports, `sub_records.reduce_scatter_plan` = the reduce-scatter / all-gather plan beside
# the default (one all-reduce per mode below 1 MiB).
set -u
OUT=${1:-gpurun_out/scale}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
PORT=29611
for WL in cp4_s200_r10 cp4_s400_r20; do
  : > "$OUT/scale_$WL.jsonl"
  for N in 1 2 4 8; do
    if [ "$N" -eq 1 ]; then
      python bench.py --gpus 1 --steps 20 --warmup 5 --workload $WL --no-cpu-baseline --no-config-records \
        | tee -a "$OUT/scale_$WL.jsonl"
    else
      python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
        --master-port $PORT bench.py --gpus $N --steps 20 --warmup 5 --workload $WL --no-cpu-baseline \
        | tee -a "$OUT/scale_$WL.jsonl"
      PORT=$((PORT + 1))
    fi
  done
done
python - "$OUT" <<'PY'
import json, sys, os
out = sys.argv[1]
for wl in ("cp4_s200_r10", "cp4_s400_r20"):
    rows = [json.loads(l) for l in open(os.path.join(out, f"scale_{wl}.jsonl")) if l.startswith("{")]
    if not rows:
        continue
    base = rows[0]["value"]
    print(f"{wl}: N, sweeps/s, speed-up, efficiency, rccl_ranks, reduce-scatter plan sweeps/s")
    for r in rows:
        rs = r.get("sub_records", {}).get("reduce_scatter_plan", {}).get("value")
        print(f"  {r['n_gpus']}, {r['value']:.1f}, {r['value'] / base:.2f}, "
              f"{r['value'] / base / r['n_gpus']:.2f}, {r.get('rccl_ranks')}, {rs}")
PY
