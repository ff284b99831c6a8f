#!/bin/bash
# Round 5, session H: gemmDecRows with a deeper ring of k-steps in flight (A/B against two), and the N > 1 code path of bench.py (two ranks on one GPU through gloo).
out=gpurun_out/${1:-r5h}; mkdir -p $out; export TMPDIR=/tmp
timeout 300 python -c "from whisper_amd import canary; canary.run_all()" 2>&1 | tail -1 | tee $out/canary.log
grep -q "mel ok" $out/canary.log || { echo "CANARY FAILED"; exit 3; }
echo "== tests"; date
timeout 900 python -m pytest tests/test_big_batch.py -q -rP -k "mul_mat or toy_model" > $out/test.log 2>&1; echo "pytest rc=$?" | tee -a $out/test.log
grep -E "passed|failed|FAILED|Error" $out/test.log | tail -5
echo "== options 224"; date
SWEEP_REPS=3 timeout 600 python tools/r5_sweep.py options 32 "default;dec_depth=2;default;dec_depth=2" > $out/options224.log 2>&1; echo "rc=$?"; grep -v "^\[" $out/options224.log | grep -E "lock-step batch|kernel table|gemvFused" | head -20
echo "== dry run, two ranks"; date
bash tools/dry_run_2ranks.sh ${1:-r5h}
date
