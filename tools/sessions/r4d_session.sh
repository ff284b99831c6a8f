#!/bin/bash
# Round 4, session D: counters that settle the clock / MFMA-busy question for gemmTiled8, the vendor library on the same shapes, groupM A/B,
# and a re-run of the tests that touch the rebuilt mailbox / per-device context count.
out=gpurun_out/r4D; mkdir -p $out; export TMPDIR=/tmp
timeout 300 python -c "from whisper_amd import canary; canary.run_all()" 2>&1 | tail -2
echo "== hipBLASLt yardstick"; timeout 300 python tools/hipblaslt_ref.py 2>&1 | tee $out/hipblaslt.txt
echo "== groupM"; for g in 2 4 6 8; do echo "groupM $g"; WH_GEMM_GROUP_M=$g PROBE_VARIANTS=40 PROBE_ROUNDS=3 PROBE_SHAPES=168000x4096x1024,168000x1024x1024,168000x1024x4096,168000x3072x1024 timeout 200 python tools/gemm8_probe.py; done 2>&1 | tee $out/groupm.txt
echo "== pmc"; timeout 900 python tools/pmc_gemm.py run $out/pmc 2>&1 | tail -30
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_model.py tests/test_host_api.py tests/test_batch_api.py tests/test_cli.py -m gpu -q -x 2>&1 | tail -5
