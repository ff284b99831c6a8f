#!/bin/bash
# Round 5, session C: the wave self-attention kernel (op test + A/B in the model), beam5 batch x in-flight combinations and its kernel table.
out=gpurun_out/${1:-r5c}; mkdir -p $out; export TMPDIR=/tmp
timeout 300 python -c "from whisper_amd import canary; canary.run_all()" 2>&1 | tail -2 | tee $out/canary.log
grep -q "mel ok" $out/canary.log || { echo "CANARY FAILED"; exit 3; }
echo "== tests"; date
timeout 900 python -m pytest tests/test_big_batch.py -q -rP -k "self_attention or toy_model or chain_medium" > $out/test.log 2>&1; echo "pytest rc=$?" | tee -a $out/test.log
grep -E "passed|failed|FAILED|Error|wave kernel vs|leave the small|lock step vs" $out/test.log | tail -30
echo "== options 224"; date
timeout 600 python tools/r5_sweep.py options 32 "default;self_fuse_max_rows=128;self_fuse_max_rows=128,dec_tile=42;self_fuse_max_rows=128,self_wave_min_rows=100000" > $out/options224.log 2>&1; echo "rc=$?"; grep -v "^\[" $out/options224.log | grep -E "lock-step batch|kernel table|selfBlockDec|gemvFused|attentionDec |layerNormDec" | head -40
echo "== options 448"; date
timeout 600 python tools/r5_sweep.py options 64 "default;self_fuse_max_rows=128" > $out/options448.log 2>&1; echo "rc=$?"; grep -v "^\[" $out/options448.log | grep -E "lock-step batch|kernel table|selfBlockDec|gemvFused|attentionDec |layerNormDec" | head -20
echo "== beam5 combinations"; date
for cfg in "8 1" "4 2" "2 4" "1 8"; do set -- $cfg
  timeout 300 python bench.py --workload beam5 --steps 8 --warmup 2 --batch $1 --inflight $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('beam5 windows per batch $1, in flight $2:', d['value'], 'audio-s/s', d['sequences_per_second'], 'seq/s', d['ms_per_step'], 'ms', d['tokens_checksum'])"
done
timeout 300 python bench.py --workload beam5 --steps 8 --warmup 2 --batch 8 --inflight 1 --beam-host 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('beam5 HOST-ranked 8 x 1:', d['value'], 'audio-s/s', d['ms_per_step'], 'ms', d['tokens_checksum'])"
echo "== beam kernel table"; date
timeout 300 python tools/beam_prof.py > $out/beam_prof.log 2>&1; grep -v "^\[" $out/beam_prof.log | tail -22
date
