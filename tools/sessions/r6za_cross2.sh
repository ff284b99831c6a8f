#!/bin/bash
# Round 6, session ZA: attentionDecM with the residual rows and all V tiles requested early; ablations; the whole GPU suite on this build; beam5
out=gpurun_out/${1:-r6za}; mkdir -p $out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 600 python tools/cross_time.py > $out/cross_time.txt 2>&1; tail -6 $out/cross_time.txt
timeout 600 python tools/cross_ablate.py > $out/cross_ablate.txt 2>&1; tail -10 $out/cross_ablate.txt
timeout 600 python bench.py --workload beam5 --model large-v2 --no-cpu-baseline 2>$out/beam.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('beam5', d['value'], d['ms_per_step'], d.get('tokens_checksum'))"
timeout 1500 python -m pytest tests -m gpu -q > $out/test.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED|^ERROR" $out/test.log | tail -12
