#!/bin/bash
tag=${1:-s12}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== tests (ops + model)" ; date
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -rP > $out/test.log 2>&1 ; echo "pytest rc=$?" | tee -a $out/test.log
grep -E "passed|failed|FAILED|Error" $out/test.log | tail -5
grep -E "decode rows split-K|gelu decode rows" $out/test.log | head -20
echo "== ab 2 in flight" ; date
timeout 600 python tools/ab_bench.py --rounds 2 --steps 1 --inflight 2 --windows 112 --masks default,-1048576 --kernels > $out/ab.txt 2>&1
grep -E "^mask|^   " $out/ab.txt | head -40
date
