#!/bin/bash
# Round 5, session D: self-attention as its own launches (wave kernel) also BELOW 128 sequences? Batch plans under both settings.
out=gpurun_out/${1:-r5d}; mkdir -p $out; export TMPDIR=/tmp
timeout 300 python -c "from whisper_amd import canary; canary.run_all()" 2>&1 | tail -1 | tee $out/canary.log
grep -q "mel ok" $out/canary.log || { echo "CANARY FAILED"; exit 3; }
P="16x2:20,16x2:32,64x2:64,64x2:128"
echo "== plans, default (fused self block up to 128 sequences)"; date
SWEEP_REPS=3 timeout 600 python tools/r5_sweep.py plans "$P" 2>/dev/null | grep "audio-s/s" | tee $out/plans_default.log
echo "== plans, self-attention as its own launches from 9 sequences up"; date
WH_OPT_SELF_FUSE_MAX_ROWS=8 WH_OPT_SELF_WAVE_MIN_ROWS=8 SWEEP_REPS=3 timeout 600 python tools/r5_sweep.py plans "$P" 2>/dev/null | grep "audio-s/s" | tee $out/plans_unfused.log
echo "== plans, default again (box drift check)"; date
SWEEP_REPS=3 timeout 600 python tools/r5_sweep.py plans "16x2:20,16x2:32" 2>/dev/null | grep "audio-s/s" | tee $out/plans_default2.log
echo "== large-v2, both"; date
SWEEP_MODEL=large-v2 SWEEP_REPS=2 timeout 600 python tools/r5_sweep.py plans "16x2:20,64x2:64" 2>/dev/null | grep "audio-s/s" | tee $out/plans_large_default.log
WH_OPT_SELF_FUSE_MAX_ROWS=8 WH_OPT_SELF_WAVE_MIN_ROWS=8 SWEEP_MODEL=large-v2 SWEEP_REPS=2 timeout 600 python tools/r5_sweep.py plans "16x2:20,64x2:64" 2>/dev/null | grep "audio-s/s" | tee $out/plans_large_unfused.log
date
