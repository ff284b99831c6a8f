#!/bin/bash
# Round 6, session O: gemm_mf16 = 1 as the default -- GPU suite, smoke(), the default bench line.
out=gpurun_out/${1:-r6o}; mkdir -p $out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $out/test.log 2>&1; echo "pytest rc=$?" | tee -a $out/test.log
grep -E "passed|failed|FAILED|^ERROR" $out/test.log | tail -8
timeout 300 python -c "import __graft_entry__ as e; e.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/smoke.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], d['value_r04_definition'], r['frac'], r['mfma_kernel']['frac'], r['hbm_kernel']['frac'], r['end_to_end']['frac'], d['single_stream']['value'], d['large_v2']['value'])"
