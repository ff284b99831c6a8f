#!/bin/bash
# Round 6, session D: does leaving the persistent product fewer CUs pay with two contexts in flight, now that the pair probe shows the product loses only
# 14 % on 160 CUs (profiles/r06_evidence/pair_probe.txt)?  WH_GEMM_SPARE_CUS = 32 (default) / 64 / 96 / 128 at the driver's batch geometry.
out=gpurun_out/${1:-r6d}; mkdir -p $out; export TMPDIR=/tmp
F="--no-roofline --no-cpu-baseline --no-single-stream --no-large --no-boundary --no-ids-check --no-small-job --no-beam"
for spare in 32 64 96 128 32; do
  WH_GEMM_SPARE_CUS=$spare timeout 200 python bench.py --steps 8 --warmup 2 $F 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('spare CUs $spare:', d['value'], 'audio-s/s', d['ms_per_step'], 'ms per step')" | tee -a $out/spare.log
done
date
