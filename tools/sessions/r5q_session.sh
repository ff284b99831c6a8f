#!/bin/bash
# Round 5, session Q: batch sizes and the cross-attention's rounds of 512 workgroups (windows x 16 heads at 2 workgroups per CU): 70 windows = 2.19 rounds, 63 = 1.97.
out=gpurun_out/${1:-r5q}; mkdir -p $out; export TMPDIR=/tmp
timeout 300 python -c "from whisper_amd import canary; canary.run_all()" 2>&1 | tail -1 | tee $out/canary.log
grep -q "mel ok" $out/canary.log || { echo "CANARY FAILED"; exit 3; }
SWEEP_REPS=3 timeout 900 python tools/r5_sweep.py plans "p10+10:2,p9+11:2,p11+9:2,p13+7:2,p9+9+2:3,p10+10:2,p16+16:2,p18+14:2,p9+23:2,p13+19:2,p16+16:2" 2>/dev/null | grep "audio-s/s" | tee $out/plans.log
date
