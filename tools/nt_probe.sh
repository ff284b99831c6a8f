B="timeout 300 python bench.py --no-roofline --no-cpu-baseline --no-single-stream --no-large"
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
D=$(python -c "from whisper_amd import binding as b; print(b.TUNE_DEFAULT)")
N=$(python -c "from whisper_amd import binding as b; print(b.TUNE_DEFAULT | b.TUNE_ATTN_DEC_NT)")
WH_TUNING=$D $B 2>/dev/null | pick default
WH_TUNING=$N $B 2>/dev/null | pick nt
WH_TUNING=$D $B 2>/dev/null | pick default
WH_TUNING=$N $B 2>/dev/null | pick nt
