#!/bin/bash
# Round 6, session L: where the encoder product's K loop stands against the guide's 8-phase template figures (4096^3 / 8192^3) and the vendor
# library on the same box -- a K sweep at the encoder's M and N gives time per K tile (a) and per output tile (b) for all three.
out=gpurun_out/${1:-r6l}; mkdir -p $out; export TMPDIR=/tmp
S="4096x4096x4096,8192x8192x8192,168000x1024x1024,168000x1024x2048,168000x1024x4096,168000x1024x8192,168000x4096x1024,168000x4096x4096"
PROBE_SHAPES=$S PROBE_VARIANTS=40,50 timeout 600 python tools/gemm8_probe.py > $out/ours.txt 2>&1; echo "ours rc=$?"
PROBE_SHAPES=$S timeout 600 python tools/hipblaslt_ref.py > $out/vendor.txt 2>&1; echo "vendor rc=$?"
cat $out/ours.txt $out/vendor.txt
timeout 300 whisper_amd/lib/mfma-power-probe > $out/power_probe.txt 2>&1; echo "power probe rc=$?"; cat $out/power_probe.txt
