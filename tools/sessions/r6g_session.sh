#!/bin/bash
# Round 6, session G: the whole GPU suite on the round's state so far (exact mode, attentionEncW as the timed encoder attention), then the default bench line.
out=gpurun_out/${1:-r6g}; mkdir -p $out; export TMPDIR=/tmp
timeout 300 python -c "from whisper_amd import canary; canary.run_all()" 2>&1 | tail -2
date
timeout 2400 python -m pytest tests -m gpu -q -rP --durations=15 > $out/test.log 2>&1; echo "pytest rc=$?" | tee -a $out/test.log
grep -E "passed|failed|FAILED|^ERROR" $out/test.log | tail -15
date
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['unit'], d['ms_per_step'], 'small_job', d.get('small_job',{}).get('value'), 'roofline', d['roofline'].get('frac'), d['roofline'].get('which'))"
date
