#!/bin/bash
# enc_chunk sweep on the final kernels (windows per encoder pass; 448-window batches)
for v in 128 64 224 32 128; do
  echo "WH_OPT_ENC_CHUNK=$v"
  WH_OPT_ENC_CHUNK=$v timeout 600 python bench.py --no-cpu-baseline --no-single-stream --no-large --no-boundary --no-workloads --no-small-job --no-ids-check 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('bench', d['value'], d['ms_per_step'], 'mfma', r['mfma_kernel']['frac'], r['mfma_kernel']['avg_launch_us'], 'enc', r['encoder_attention']['avg_launch_us'], 'ln', d['kernels']['layerNorm'])"
done
