#!/bin/bash
# Round 5, session A: the big lock-step batches -- new parity tests, batch-plan sweep, decode-option A/B with kernel tables.
out=gpurun_out/${1:-r5a}; mkdir -p $out; export TMPDIR=/tmp
timeout 300 python -c "from whisper_amd import canary; canary.run_all()" 2>&1 | tee $out/canary.log
grep -q "mel ok" $out/canary.log || { echo "CANARY FAILED"; exit 3; }
echo "== new tests"; date
timeout 1200 python -m pytest tests/test_big_batch.py -q -rP -x > $out/test_big.log 2>&1; echo "pytest rc=$?" | tee -a $out/test_big.log
grep -E "passed|failed|FAILED|Error|PARITY MODE|lock step vs|leave the small|rows \[0, 112\)" $out/test_big.log | tail -40
echo "== plans"; date
timeout 600 python tools/r5_sweep.py plans "${PLANS:-16x2:32,64x1:32,64x1:64,32x2:64,64x1:20,16x2:20}" > $out/plans.log 2>&1; echo "plans rc=$?"; grep "audio-s/s" $out/plans.log
echo "== options"; date
timeout 600 python tools/r5_sweep.py options ${OPT_CLIPS:-32} "${OPTS:-default;dec_tile=1;dec_tile=42;self_nq=4;self_nq=8;self_fuse_max_rows=128;vocab_decrows=1}" > $out/options.log 2>&1; echo "options rc=$?"; grep -v "^\[" $out/options.log | head -120
date
