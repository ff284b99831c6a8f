#!/bin/bash
out=gpurun_out/s17
mkdir -p $out
for w in 1 7; do
echo "== lone stream, $w window(s): default | no fused self block | no fused cross query | neither"
timeout 300 python tools/ab_bench.py --rounds 2 --steps 1 --inflight 1 --windows $w --masks default,-4096,-1024,-5120 > $out/abf_w$w.txt 2>&1
grep -E "^mask" $out/abf_w$w.txt | head -14
done
