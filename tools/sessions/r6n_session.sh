#!/bin/bash
# Round 6, session N: gemmTiled8 with v_mfma_f32_16x16x32_f16 (option gemm_mf16) -- bits against the 32x32x16 kernel, the probe, the model-level identity test, bench A/B.
out=gpurun_out/${1:-r6n}; mkdir -p $out; export TMPDIR=/tmp
WH_PROBE_REF=40 PROBE_ROUNDS=3 PROBE_SHAPES=168000x1024x1024,168000x3072x1024,168000x4096x1024,168000x1024x4096 PROBE_VARIANTS=40,52 timeout 500 python tools/gemm8_probe.py 2>&1 | grep -v amdgpu.ids | tee $out/probe.txt
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "bit_identical_under_the_gemm_variants or medium_shape or encoder" 2>&1 | tail -5 | tee $out/test_subset.txt
for v in 0 1 0 1; do
  echo "WH_OPT_GEMM_MF16=$v"
  WH_OPT_GEMM_MF16=$v timeout 600 python bench.py --no-cpu-baseline --no-single-stream --no-large --no-boundary --no-workloads --no-small-job 2>$out/bench_mf16_$v.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('bench', d['value'], d['ms_per_step'], 'mfma', r['mfma_kernel']['frac'], r['mfma_kernel'].get('us_per_launch'), 'timed_ids', d['parity'].get('timed_ids') if d.get('parity') else None)"
done 2>&1 | tee $out/bench_ab.txt
