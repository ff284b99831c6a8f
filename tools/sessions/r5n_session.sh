#!/bin/bash
# Round 5, session N: the MLP down-projection of 33..128 rows with all rows per workgroup and 8 waves over K (option dec_deep_rows): tests, then plans off / on.
out=gpurun_out/${1:-r5n}; mkdir -p $out; export TMPDIR=/tmp
timeout 300 python -c "from whisper_amd import canary; canary.run_all()" 2>&1 | tail -1 | tee $out/canary.log
grep -q "mel ok" $out/canary.log || { echo "CANARY FAILED"; exit 3; }
echo "== tests"; date
timeout 900 python -m pytest tests/test_big_batch.py -q -rP -k "deep_decode or wide_" > $out/test.log 2>&1; echo "pytest rc=$?" | tee -a $out/test.log
grep -E "passed|failed|FAILED|Error|one-tile vs" $out/test.log | tail -12
for on in 0 1 0 1; do
  echo "== plans, dec_deep_rows=$on"; date
  WH_OPT_DEC_DEEP_ROWS=$on SWEEP_REPS=3 timeout 600 python tools/r5_sweep.py plans "16x2:20,16x2:32" 2>/dev/null | grep "audio-s/s" | tee -a $out/plans_$on.log
done
echo "== kernel tables at 70 windows"; date
timeout 600 python tools/r5_sweep.py options 10 "default;dec_deep_rows=1" > $out/options70.log 2>&1; grep -v "^\[" $out/options70.log | grep -E "lock-step batch|kernel table|gemvFused" | head
date
