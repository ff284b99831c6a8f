#!/bin/bash
# Round 5, session W (the last GPU seconds of the round): smoke() on the final commit, then two data points for the next round -- three contexts in flight
# at 42 / 48 clips per step against the default (64 clips, two contexts), the driver's 20 steps.
out=gpurun_out/${1:-r5w}; mkdir -p $out; export TMPDIR=/tmp
timeout 60 python -c "import __graft_entry__ as e; e.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/smoke.log
F="--no-roofline --no-cpu-baseline --no-single-stream --no-large --no-boundary --no-ids-check --no-small-job"
for cfg in "48 3" "42 3"; do set -- $cfg
  timeout 70 python bench.py --steps 20 --warmup 3 --clips-per-step $1 --inflight $2 $F 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('clips per step $1, contexts in flight $2:', d['value'], 'audio-s/s', d['ms_per_step'], 'ms per step')" | tee -a $out/inflight3.log
done
date
