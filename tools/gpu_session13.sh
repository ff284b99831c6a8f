#!/bin/bash
tag=${1:-s13}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== tests (ops + model)" ; date
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -rP > $out/test.log 2>&1 ; echo "pytest rc=$?" | tee -a $out/test.log
grep -E "passed|failed|FAILED|Error" $out/test.log | tail -5
grep -E "on vs off" $out/test.log | head -40
echo "== gemv time" ; date
timeout 200 python tools/gemv_time.py 112 2>&1 | grep "M="
echo "== ab 2 in flight" ; date
timeout 600 python tools/ab_bench.py --rounds 2 --steps 1 --inflight 2 --windows 112 --masks default,-4194304,-3145728 --kernels > $out/ab.txt 2>&1
grep -E "^mask|^   " $out/ab.txt | head -40
date
