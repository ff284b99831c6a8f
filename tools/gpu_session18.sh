#!/bin/bash
out=gpurun_out/s18
mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -x -rP -k "fused_launches or batch_invariance" > $out/test.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|FAILED|Error" $out/test.log | tail -3; grep -E "NQ8 on vs off" $out/test.log
timeout 600 python tools/ab_bench.py --rounds 2 --steps 1 --inflight 2 --windows 112 --masks default,-16777216 --kernels > $out/ab.txt 2>&1
grep -E "^mask|^   (selfBlock|gemvFused|attentionDecCross)" $out/ab.txt | head -20
