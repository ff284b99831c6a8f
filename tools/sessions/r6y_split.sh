#!/bin/bash
# Round 6, session Y: the K-split MLP down-projection of 33 .. 128 rows (dec_split): isolated times, tests, beam5 / shard256 / small-job A/B
out=gpurun_out/${1:-r6y}; mkdir -p $out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 600 python tools/deep_time.py > $out/deep_time.txt 2>&1; tail -12 $out/deep_time.txt
timeout 900 python -m pytest tests/test_big_batch.py -m gpu -q -x > $out/test_big.log 2>&1; tail -3 $out/test_big.log
for v in 0 1 0 1; do
  echo "WH_OPT_DEC_SPLIT=$v"
  WH_OPT_DEC_SPLIT=$v timeout 600 python bench.py --workload beam5 --model large-v2 --no-cpu-baseline 2>$out/beam_$v.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('beam5', d['value'], d['ms_per_step'])"
done
for v in 0 1; do
  WH_OPT_DEC_SPLIT=$v timeout 600 python bench.py --workload shard256 --model large-v2 --no-cpu-baseline 2>$out/shard_$v.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shard256', d['value'], d['ms_per_step'])"
done
