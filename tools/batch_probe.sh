B="timeout 300 python bench.py --no-roofline --no-cpu-baseline --no-single-stream --no-large"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['config'].get('batch_plan'))"; }
$B --steps 20 --warmup 5 2>/dev/null | pick k20_f2
$B --steps 20 --warmup 5 --inflight 3 2>/dev/null | pick k20_f3
$B --steps 32 --warmup 1 --inflight 3 2>/dev/null | pick k32_f3
$B --steps 32 --warmup 1 2>/dev/null | pick k32_f2
