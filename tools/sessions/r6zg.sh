#!/bin/bash
# Round 6, session ZG: the ranked beam step's cache reorder in one launch through registers (reorder_group): beam tests, beam5 A B A B, kernel statistics
out=gpurun_out/${1:-r6zg}; mkdir -p $out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -x -k "beam" > $out/test_beam.log 2>&1; tail -3 $out/test_beam.log
for v in 0 1 0 1; do
  echo "WH_OPT_REORDER_GROUP=$v"
  WH_OPT_REORDER_GROUP=$v timeout 600 python bench.py --workload beam5 --model large-v2 --no-cpu-baseline 2>$out/beam_$v.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('beam5', d['value'], d['ms_per_step'], d.get('tokens_checksum'))"
done
rm -rf /tmp/prof_beam
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_beam -- python $R/bench.py --workload beam5 --model large-v2 --no-cpu-baseline --steps 2 --warmup 1 > $R/$out/beam_prof.json 2> $R/$out/beam_prof.err
cd $R
f=$(find /tmp/prof_beam -name "*kernel_stats.csv" | head -1); cp $f $out/beam_kernel_stats.csv 2>/dev/null
grep -E "reorder" $out/beam_kernel_stats.csv | cut -c1-200
