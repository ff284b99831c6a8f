#!/bin/bash
# Round 6, session ZC: vocabulary product of 33 .. 64 rows on gemmDecTile (vocab_lds), the beam ranking's logarithms on their own lanes; beam5 A/B and the beam tests
out=gpurun_out/${1:-r6zc}; mkdir -p $out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 600 python tools/vocab_time.py > $out/vocab_time.txt 2>&1; tail -13 $out/vocab_time.txt
for v in 0 1 0 1; do
  echo "WH_OPT_VOCAB_LDS=$v"
  WH_OPT_VOCAB_LDS=$v timeout 600 python bench.py --workload beam5 --model large-v2 --no-cpu-baseline 2>$out/beam_$v.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('beam5', d['value'], d['ms_per_step'], d.get('tokens_checksum'))"
done
timeout 1200 python -m pytest tests -m gpu -q -x -k "beam or vocab or wide or deep" > $out/test_beam.log 2>&1; tail -3 $out/test_beam.log
