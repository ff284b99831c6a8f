B="timeout 300 python bench.py --no-roofline --no-cpu-baseline --no-single-stream --no-large"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['config'].get('batch_plan'))"; }
$B --steps 20 --warmup 2 --plan 16,4 2>/dev/null | pick k20_16_4
$B --steps 20 --warmup 2 2>/dev/null | pick k20_default
$B --steps 16 --warmup 2 --plan 16 2>/dev/null | pick k16_16
$B --steps 16 --warmup 2 2>/dev/null | pick k16_default
$B --steps 40 --warmup 2 --plan 16,16,8 2>/dev/null | pick k40_16_16_8
$B --steps 40 --warmup 2 2>/dev/null | pick k40_default
$B --steps 40 --warmup 2 --plan 14,13,13 2>/dev/null | pick k40_14_13_13
