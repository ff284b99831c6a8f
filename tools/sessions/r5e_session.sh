#!/bin/bash
# Round 5, session E: one context or two at each K with the final decode path.
out=gpurun_out/${1:-r5e}; mkdir -p $out; export TMPDIR=/tmp
timeout 300 python -c "from whisper_amd import canary; canary.run_all()" 2>&1 | tail -1 | tee $out/canary.log
grep -q "mel ok" $out/canary.log || { echo "CANARY FAILED"; exit 3; }
SWEEP_REPS=3 timeout 900 python tools/r5_sweep.py plans "64x1:20,16x2:20,7x3:20,64x1:32,16x2:32,11x3:32,64x1:64,64x2:64,22x3:64,64x3:192" 2>/dev/null | grep "audio-s/s" | tee $out/plans.log
date
