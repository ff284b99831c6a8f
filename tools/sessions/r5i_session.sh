#!/bin/bash
# Round 5, session I: does an encoder product that leaves room on the CU let the other context's decode run under it? The existing non-persistent kernels as stand-ins:
# 16-wave 256x256x64 (128 KiB LDS, one workgroup per CU, not persistent) and 128x128x32 (3 workgroups per CU, 40 KiB each) against the persistent gemmTiled8.
out=gpurun_out/${1:-r5i}; mkdir -p $out; export TMPDIR=/tmp
timeout 300 python -c "from whisper_amd import canary; canary.run_all()" 2>&1 | tail -1 | tee $out/canary.log
grep -q "mel ok" $out/canary.log || { echo "CANARY FAILED"; exit 3; }
D=$(python -c "from whisper_amd import binding; print(binding.TUNE_DEFAULT)")
A=$(python -c "from whisper_amd import binding; print(binding.TUNE_DEFAULT & ~16)")
B=$(python -c "from whisper_amd import binding; print(binding.TUNE_DEFAULT & ~16 & ~8)")
for cfg in "default $D" "16wave-nonpersistent $A" "128x128x32 $B" "default $D"; do set -- $cfg
  echo "== $1 (WH_TUNING=$2)"
  WH_TUNING=$2 SWEEP_REPS=3 timeout 600 python tools/r5_sweep.py plans "64x2:64,64x1:32,16x2:20" 2>/dev/null | grep "audio-s/s" | tee -a $out/plans_$1.log
done
date
