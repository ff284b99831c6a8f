#!/bin/bash
tag=${1:-s2}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
echo "== tests" ; date
timeout 1200 python -m pytest tests -m gpu -q -s > $out/test.log 2>&1 ; echo "pytest rc=$?" | tee -a $out/test.log
grep -E "passed|failed|FAILED|Error" $out/test.log | tail -30
echo "== rocprof ab" ; date
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -- python $GRAFT_REPO_ROOT/tools/ab_bench.py --rounds 1 --steps 1 --masks default,-512 --kernels > $GRAFT_REPO_ROOT/$out/ab_prof.txt 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_ab -name "*kernel_stats.csv" | head -1); cp $f $out/ab_kernel_stats.csv 2>/dev/null
head -40 $out/ab_kernel_stats.csv | cut -c1-200
tail -30 $out/ab_prof.txt
date
