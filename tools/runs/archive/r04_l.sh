# round 4, call l: the driver's bench command with the new records; driver / rccl / fixture tests
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tools/gpu_steps.sh \
 "r04l_bench|600|python bench.py --gpus 1 --steps 20 --warmup 5 | tail -1 > gpurun_out/r04l_bench.json; python3 -c \"import json; d=json.load(open('gpurun_out/r04l_bench.json')); print(d['value'], d['roofline']['frac']); print(json.dumps(d['placement'])[:1500]); s=d['sub_records']; print({k:(v.get('value'), (v.get('roofline') or {}).get('frac')) for k,v in s.items() if isinstance(v,dict)}); print(json.dumps(s['cfg3_pp'])[:1800]); print(json.dumps(s['cfg5_tucker'])[:900]); print(json.dumps(s['shard_probe']))\"" \
 "r04l_tests|900|python -m pytest tests/test_gpu_driver.py tests/test_gpu_rccl.py tests/test_ctf_fixtures.py -x -q"
