B="timeout 300 python bench.py --no-roofline --no-cpu-baseline --no-single-stream --no-large"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for n in 0 8 16 32 64; do WH_GEMM_SPARE_CUS=$n $B 2>/dev/null | pick spare$n; done
WH_GEMM_SPARE_CUS=16 $B --inflight 3 2>/dev/null | pick spare16x3
