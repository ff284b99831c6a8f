#!/bin/bash
tag=${1:-s4}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
echo "== tests (host api + fused)" ; date
timeout 600 python -m pytest tests/test_host_api.py tests/test_cli.py tests/test_gpu_model.py -m gpu -q -x -k "host or cli or fused or hypoth or run_full" > $out/test.log 2>&1 ; echo "pytest rc=$?" | tee -a $out/test.log
grep -E "passed|failed|FAILED|Error" $out/test.log | tail -10
echo "== sweep" ; date
timeout 900 python tools/shape_sweep.py "4x3,8x2,8x3,12x2,16x1,16x2" --split > $out/sweep.txt 2>&1
grep -E "clips/batch" $out/sweep.txt
date
