#!/bin/bash
# non-temporal stores / residual loads in the encoder product's lean epilogue (option gemm_nt): bench A/B
for v in 0 1 2 0 1 2; do
  echo "WH_OPT_GEMM_NT=$v"
  WH_OPT_GEMM_NT=$v timeout 600 python bench.py --no-cpu-baseline --no-single-stream --no-large --no-boundary --no-workloads --no-small-job --no-ids-check 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('bench', d['value'], d['ms_per_step'], 'mfma', r['mfma_kernel']['frac'], r['mfma_kernel']['avg_launch_us'], 'hbm', r['hbm_kernel']['frac'], 'enc', r['encoder_attention']['avg_launch_us'], 'ln', d['kernels']['layerNorm']['avg_us'], d['tokens_checksum'])"
done
