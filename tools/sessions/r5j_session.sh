#!/bin/bash
# Round 5, session J: the same stand-ins with the encoders of the contexts in flight one at a time (TUNE_ENC_SERIAL: decode of one context under the encoder of the other),
# four batches of 32 clips so that the phases can interleave.
out=gpurun_out/${1:-r5j}; mkdir -p $out; export TMPDIR=/tmp
timeout 300 python -c "from whisper_amd import canary; canary.run_all()" 2>&1 | tail -1 | tee $out/canary.log
grep -q "mel ok" $out/canary.log || { echo "CANARY FAILED"; exit 3; }
D=$(python -c "from whisper_amd import binding; print(binding.TUNE_DEFAULT)")
DS=$(python -c "from whisper_amd import binding; print(binding.TUNE_DEFAULT | 134217728)")
B=$(python -c "from whisper_amd import binding; print(binding.TUNE_DEFAULT & ~16 & ~8)")
BS=$(python -c "from whisper_amd import binding; print((binding.TUNE_DEFAULT & ~16 & ~8) | 134217728)")
for cfg in "default $D" "default+serial $DS" "128x128x32 $B" "128x128x32+serial $BS"; do set -- $cfg
  echo "== $1 (WH_TUNING=$2)"
  WH_TUNING=$2 SWEEP_REPS=2 timeout 600 python tools/r5_sweep.py plans "32x2:128,32x3:192" 2>/dev/null | grep "audio-s/s" | tee -a $out/plans_$1.log
done
date
