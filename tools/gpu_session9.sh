#!/bin/bash
tag=${1:-s9}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== tests (ops + model)" ; date
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x > $out/test.log 2>&1 ; echo "pytest rc=$?" | tee -a $out/test.log
grep -E "passed|failed|FAILED|Error" $out/test.log | tail -5
echo "== ab" ; date
timeout 500 python tools/ab_bench.py --rounds 2 --steps 1 --inflight 2 --windows 112 --masks default,-65536 --kernels > $out/ab.txt 2>&1
grep -E "^mask|^   " $out/ab.txt | head -30
echo "== rocprof bench (timed region only)" ; date
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $R/bench.py --no-roofline --no-cpu-baseline --no-single-stream --no-large > $R/$out/bench_prof.json 2> $R/$out/bench_prof.err
cd $R
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1); cp $f $out/bench_kernel_stats.csv 2>/dev/null
head -16 $out/bench_kernel_stats.csv | cut -c1-200
head -c 300 $out/bench_prof.json
date
