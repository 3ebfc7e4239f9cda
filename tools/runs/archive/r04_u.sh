cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tools/gpu_steps.sh \
 "r04u_tests|1100|python -m pytest tests -m gpu -x -q --durations=8" \
 "r04u_bench|600|python bench.py --gpus 1 --steps 20 --warmup 5 | tail -1 > gpurun_out/r04u_bench.json; python3 -c \"import json; d=json.load(open('gpurun_out/r04u_bench.json')); print(d['value'], d['roofline']['frac']); s=d['sub_records']; print({k:(v.get('value'), (v.get('roofline') or {}).get('frac')) for k,v in s.items() if isinstance(v,dict)}); print(json.dumps(s['cfg5_tucker'])[:700])\""
