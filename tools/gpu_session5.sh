#!/bin/bash
tag=${1:-s5}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
echo "== tests" ; date
timeout 900 python -m pytest tests -m gpu -q -s -x > $out/test.log 2>&1 ; echo "pytest rc=$?" | tee -a $out/test.log
grep -E "passed|failed|FAILED|Error" $out/test.log | tail -10
grep -E "flash_attention" $out/test.log | cut -c1-140 | head -30
echo "== ab" ; date
timeout 500 python tools/ab_bench.py --rounds 2 --steps 2 --masks default,-65536 --kernels > $out/ab.txt 2>&1
grep -E "^mask|^   " $out/ab.txt
echo "== ab inflight 2 x 56" ; date
timeout 500 python tools/ab_bench.py --rounds 2 --steps 2 --inflight 2 --windows 56 --masks default,-65536 > $out/ab3.txt 2>&1
grep -E "^mask" $out/ab3.txt
date
