B="timeout 300 python bench.py --no-roofline --no-cpu-baseline --no-single-stream --no-large"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
$B 2>/dev/null | pick base
WH_TUNING=33546234 $B 2>/dev/null | pick prio
for n in 96 128 160 192; do WH_TUNING=33546234 WH_ENC_CUS=$n $B 2>/dev/null | pick enc$n; done
WH_TUNING=33546234 WH_ENC_CUS=160 $B --inflight 3 2>/dev/null | pick enc160x3
$B --inflight 3 2>/dev/null | pick basex3
