#!/bin/bash
# One gpurun call: parity tests, the bench line, probes. Everything lands under gpurun_out/$1.
tag=${1:-s1}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
echo "== tests" ; date
timeout 900 python -m pytest tests -m gpu -x -q -s > $out/test.log 2>&1 ; echo "pytest rc=$?" | tee -a $out/test.log
tail -5 $out/test.log
echo "== bench" ; date
timeout 600 python bench.py --steps 24 --warmup 2 > $out/bench.json 2> $out/bench.err ; echo "bench rc=$?"
tail -c 1500 $out/bench.json
echo "== walk probe" ; date
timeout 300 python tools/gemm_walk_probe.py > $out/gemm_walk.txt 2>&1
cat $out/gemm_walk.txt | head -40
echo "== ab" ; date
timeout 400 python tools/ab_bench.py --rounds 2 --steps 2 --masks default,-512,-1024,-2048 --kernels > $out/ab.txt 2>&1
tail -45 $out/ab.txt
date
