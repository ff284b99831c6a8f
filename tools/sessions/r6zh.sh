#!/bin/bash
# Round 6, session ZH: the persistent 256 x 256 encoder product from fewer rows (gemm_big_min_rows): beam5 (8 windows = 12000 rows) A/B
out=gpurun_out/${1:-r6zh}; mkdir -p $out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for v in 16384 8192 16384 8192 4096; do
  echo "WH_OPT_GEMM_BIG_MIN_ROWS=$v"
  WH_OPT_GEMM_BIG_MIN_ROWS=$v timeout 600 python bench.py --workload beam5 --model large-v2 --no-cpu-baseline 2>$out/beam_$v.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('beam5', d['value'], d['ms_per_step'], d.get('tokens_checksum'))"
done
