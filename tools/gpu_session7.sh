#!/bin/bash
tag=${1:-s7}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
echo "== tests" ; date
timeout 900 python -m pytest tests -m gpu -q -s -x > $out/test.log 2>&1 ; echo "pytest rc=$?" | tee -a $out/test.log
grep -E "passed|failed|FAILED|Error" $out/test.log | tail -10
echo "== ab" ; date
timeout 500 python tools/ab_bench.py --rounds 2 --steps 2 --masks default,-131072 --kernels > $out/ab.txt 2>&1
grep -E "^mask|^   (atten|gemm|self)" $out/ab.txt
echo "== ab inflight 2 x 112" ; date
timeout 500 python tools/ab_bench.py --rounds 2 --steps 1 --inflight 2 --windows 112 --masks default,-131072 > $out/ab3.txt 2>&1
grep -E "^mask" $out/ab3.txt
date
