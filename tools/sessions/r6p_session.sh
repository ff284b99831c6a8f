#!/bin/bash
# Round 6, session P: packed Q/K/V epilogue of gemmDecRows -- the big-batch tests, kernel statistics of a short bench run (one context in flight), bench line.
out=gpurun_out/${1:-r6p}; mkdir -p $out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_big_batch.py tests/test_gpu_ops.py -m gpu -q -x 2>&1 | tail -3 | tee $out/test_subset.txt
rm -rf /tmp/prof_bench
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $R/bench.py --inflight 1 --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-single-stream --no-large --no-boundary --no-workloads --no-small-job --no-ids-check > $R/$out/bench_prof.json 2> $R/$out/bench_prof.err
cd $R
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1); cp $f $out/bench_kernel_stats.csv 2>/dev/null
head -14 $out/bench_kernel_stats.csv | cut -c1-170
timeout 600 python bench.py --no-cpu-baseline --no-single-stream --no-large --no-boundary --no-workloads --no-small-job 2>$out/bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('bench', d['value'], d['ms_per_step'], 'mfma', r['mfma_kernel']['frac'], 'chain', r['decode_chain'])"
