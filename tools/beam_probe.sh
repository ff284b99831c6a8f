pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
B="timeout 300 python bench.py --workload beam5 --steps 8 --warmup 2"
$B 2>/dev/null | pick default
$B 2>/dev/null | pick default_again
WH_NO_MAILBOX=1 $B 2>/dev/null | pick nomailbox
WH_GEMM_SPARE_CUS=0 $B 2>/dev/null | pick nospare
