#!/bin/bash
tag=${1:-s14}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
echo "== ab 2 in flight: half-CU GEMM tiles" ; date
timeout 600 python tools/ab_bench.py --rounds 2 --steps 1 --inflight 2 --windows 112 --masks default,+8388608 --kernels > $out/ab.txt 2>&1
grep -E "^mask|^   (gemmTiled|attentionDecCross|attentionEnc|selfBlock|gemvFused)" $out/ab.txt | head -40
echo "== ab 3 in flight x 80 windows" ; date
timeout 600 python tools/ab_bench.py --rounds 2 --steps 1 --inflight 3 --windows 77 --masks default,+8388608 > $out/ab3.txt 2>&1
grep -E "^mask" $out/ab3.txt | head
date
