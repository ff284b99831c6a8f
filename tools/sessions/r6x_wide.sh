#!/bin/bash
# Round 6, session X: gemmDecTile with 4 / 6 / 8 row tiles for the wide products of 33 .. 128 rows (dec_lds on launchDecRowsOneTile): isolated times, the bit-identity test, beam5 A/B, kernel statistics of a beam pass
out=gpurun_out/${1:-r6x}; mkdir -p $out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 600 python tools/wide_time.py > $out/wide_time.txt 2>&1; tail -30 $out/wide_time.txt
timeout 900 python -m pytest tests/test_big_batch.py -m gpu -q -x > $out/test_big.log 2>&1; tail -3 $out/test_big.log
for v in 0 1 0 1; do
  echo "WH_OPT_DEC_LDS=$v"
  WH_OPT_DEC_LDS=$v timeout 600 python bench.py --workload beam5 --model large-v2 --no-cpu-baseline 2>$out/beam_$v.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('beam5', d['value'], d['ms_per_step'])"
done
rm -rf /tmp/prof_beam
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_beam -- python $R/bench.py --workload beam5 --model large-v2 --no-cpu-baseline --steps 2 --warmup 1 > $R/$out/beam_prof.json 2> $R/$out/beam_prof.err
cd $R
f=$(find /tmp/prof_beam -name "*kernel_stats.csv" | head -1); cp $f $out/beam_kernel_stats.csv 2>/dev/null
head -30 $out/beam_kernel_stats.csv | cut -c1-200
