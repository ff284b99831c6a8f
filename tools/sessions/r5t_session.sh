#!/bin/bash
# Round 5, session T: contexts in flight and clips per step at the driver's 20 steps.
out=gpurun_out/${1:-r5t}; mkdir -p $out; export TMPDIR=/tmp
F="--no-roofline --no-cpu-baseline --no-single-stream --no-large --no-boundary --no-ids-check --no-small-job"
for cfg in "32 2" "32 3" "32 4" "64 2" "32 2" "32 3"; do set -- $cfg
  timeout 600 python bench.py --steps 20 --warmup 5 --clips-per-step $1 --inflight $2 $F 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('clips per step $1, contexts in flight $2:', d['value'], 'audio-s/s', d['ms_per_step'], 'ms per step')" | tee -a $out/inflight.log
done
date
