#!/bin/bash
# f64 storage: which scan variant / persistence does best (A/B in one call)
run() { echo "== $*"; env "$@" python3 bench.py --dtype f64 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print(j['value'], r['avg_launch_ms'], r['frac'])"; }
run PPALS_X=1
run PPALS_SCAN_VARIANT=1
run PPALS_PERSIST_MULT=20
run PPALS_PERSIST_MULT=80
run PPALS_PERSIST_MULT=12
echo "== dt schedule"; python3 bench.py --dtype f64 --schedule dt --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print(j['value'], r['avg_launch_ms'], r['frac'])"
echo "== dt schedule variant 1"; PPALS_SCAN_VARIANT=1 python3 bench.py --dtype f64 --schedule dt --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print(j['value'], r['avg_launch_ms'], r['frac'])"
