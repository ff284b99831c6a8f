#!/bin/bash
# Round 6, session A: WH_FLAG_PARITY_EXACT on the device for the first time -- d128 first (-x), then the medium shape.
out=gpurun_out/${1:-r6a}; mkdir -p $out; export TMPDIR=/tmp
timeout 300 python -c "from whisper_amd import canary; canary.run_all()" 2>&1 | tail -3
date
timeout 1700 python -m pytest tests/test_gpu_exact.py -m gpu -x -q -rP --durations=10 > $out/exact.log 2>&1; echo "pytest rc=$?" | tee -a $out/exact.log
grep -E "passed|failed|FAILED|Error|differ|thread" $out/exact.log | tail -40
date
