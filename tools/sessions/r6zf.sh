#!/bin/bash
# Round 6, session ZF: softMaxRowsReg as the default vocabulary softmax (beam_regs), LDS-resident beam ranking: the whole GPU suite, beam5 A B A B
out=gpurun_out/${1:-r6zf}; mkdir -p $out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q > $out/test.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED|^ERROR" $out/test.log | tail -8
for v in 0 1 0 1; do
  echo "WH_OPT_BEAM_REGS=$v"
  WH_OPT_BEAM_REGS=$v timeout 600 python bench.py --workload beam5 --model large-v2 --no-cpu-baseline 2>$out/beam_$v.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('beam5', d['value'], d['ms_per_step'], d.get('tokens_checksum'))"
done
