# quick A/B helpers for a GPU session: batch-1 decode step, single stream, the short bench line
D1_STEPS=150 timeout 200 python tools/decode1_prof.py 2>&1 | grep -v amdgpu.ids | grep "greedy graph\|window loop (fetch, 1\|encode\|prompt" | tail -5
export SS_MODEL_FILE=/tmp/ss.bin; RUNS=2 timeout 200 python tools/single_stream_prof.py 2>&1 | grep "run_full" | tail -1
timeout 300 python bench.py --no-roofline --no-cpu-baseline --no-single-stream --no-large 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])"
