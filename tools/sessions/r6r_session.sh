#!/bin/bash
# Round 6, session R: gemmDecTile as the default for the decode products of 129 .. 512 rows -- GPU suite, bench A/B against dec_lds 0.
out=gpurun_out/${1:-r6r}; mkdir -p $out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $out/test.log 2>&1; echo "pytest rc=$?" | tee -a $out/test.log
grep -E "passed|failed|FAILED|^ERROR" $out/test.log | tail -8
for v in 0 1 0 1; do
  echo "WH_OPT_DEC_LDS=$v"
  WH_OPT_DEC_LDS=$v timeout 600 python bench.py --no-cpu-baseline --no-single-stream --no-large --no-boundary --no-workloads --no-small-job 2>$out/bench_$v.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('bench', d['value'], d['ms_per_step'], 'mfma', r['mfma_kernel']['frac'], 'chain ms', r['decode_chain']['ms_per_batch'], 'ids', d['parity']['timed_ids']['windows_with_identical_ids'], d['tokens_checksum'])"
done 2>&1 | tee $out/bench_ab.txt
