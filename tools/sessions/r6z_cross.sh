#!/bin/bash
# Round 6, session Z: attentionDecM (the beam step's cross-attention on the matrix cores, option cross_mfma): op tests, isolated times, beam5 A/B, the beam tests
out=gpurun_out/${1:-r6z}; mkdir -p $out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "cross_attention" -rP > $out/test_ops.log 2>&1; grep -E "matrix cores|attentionDecG against|passed|failed|Error|assert" $out/test_ops.log | tail -30
timeout 600 python tools/cross_time.py > $out/cross_time.txt 2>&1; tail -8 $out/cross_time.txt
for v in 0 1 0 1; do
  echo "WH_OPT_CROSS_MFMA=$v"
  WH_OPT_CROSS_MFMA=$v timeout 600 python bench.py --workload beam5 --model large-v2 --no-cpu-baseline 2>$out/beam_$v.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('beam5', d['value'], d['ms_per_step'], d.get('tokens_checksum'))"
done
timeout 1200 python -m pytest tests -m gpu -q -x -k "beam" > $out/test_beam.log 2>&1; tail -5 $out/test_beam.log
