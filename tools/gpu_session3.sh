#!/bin/bash
tag=${1:-s3}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
echo "== tests" ; date
timeout 1200 python -m pytest tests -m gpu -q -s -x > $out/test.log 2>&1 ; echo "pytest rc=$?" | tee -a $out/test.log
grep -E "passed|failed|FAILED|Error" $out/test.log | tail -30
grep -E " on vs off" $out/test.log | cut -c1-150
echo "== ab" ; date
timeout 500 python tools/ab_bench.py --rounds 2 --steps 2 --masks default,-4096,-8192,-16384 --kernels > $out/ab.txt 2>&1
grep -E "^mask|^   " $out/ab.txt
echo "== ab inflight 3" ; date
timeout 500 python tools/ab_bench.py --rounds 2 --steps 2 --inflight 3 --masks default,-4096 > $out/ab3.txt 2>&1
grep -E "^mask" $out/ab3.txt
date
