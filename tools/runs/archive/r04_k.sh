# round 4, call k: does the placement measurement pay on today's boxes? (interleaved A/B of the driver's command,
# config records off) + the Tucker state after the front-leaf change
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
CFG5="-model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32"
BENCH="python bench.py --gpus 1 --steps 20 --warmup 5 --no-config-records --no-cpu-baseline"
tools/gpu_steps.sh \
 "r04k_tests_tucker|600|python -m pytest tests/test_gpu_tucker.py -x -q" \
 "r04k_cfg5|100|$B/test_ALS $CFG5 -filename gpurun_out/r04k_cfg5.csv && $B/test_ALS $CFG5 -filename gpurun_out/r04k_cfg5_2.csv" \
 "r04k_ab|900|for i in 1 2 3 4; do PPALS_PLACE_TUNE=1 $BENCH | tail -1 > gpurun_out/r04k_tune1_\$i.json; PPALS_PLACE_TUNE=0 $BENCH | tail -1 > gpurun_out/r04k_tune0_\$i.json; done"
python3 - <<'PY'
import json,glob
for k in ("tune1","tune0"):
    vals=[]
    for f in sorted(glob.glob(f"gpurun_out/r04k_{k}_*.json")):
        try:
            d=json.loads(open(f).read()); vals.append((round(d["value"],1), round(d["roofline"]["frac"],3), round(d["sub_records"]["dt_schedule_f32"]["value"],1), round(d["sub_records"]["msdt_schedule_f64"]["value"],1), round(d["sub_records"]["msdt_schedule_f64"]["roofline"]["frac"],3)))
        except Exception as e: vals.append(str(e))
    print(k, vals)
PY
