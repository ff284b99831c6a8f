#!/bin/bash
tag=${1:-s21}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
# a box whose GPU faults on the first copy costs minutes per command (core-dump handlers): check once, leave at once
timeout 120 python -c "import torch; x = torch.ones(1 << 22).cuda(); print('gpu sanity', float((x * 2).sum()))" || { echo "GPU SANITY FAILED: bad box"; exit 3; }
echo "== tests" ; date
timeout 900 python -m pytest tests -m gpu -q -rP > $out/test.log 2>&1 ; echo "pytest rc=$?" | tee -a $out/test.log
grep -E "passed|failed|FAILED|Error" $out/test.log | tail -10
echo "== bench" ; date
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err ; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$out/bench.json"))
print({k:d[k] for k in ("value","ms_per_step","steps")})
print(json.dumps(d["roofline"])[:1800])
print(d["single_stream"]); print(d["large_v2"]); print(d["parity"])
PY
echo "== rocprof bench" ; date
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $R/bench.py --no-roofline --no-cpu-baseline --no-single-stream --no-large > $R/$out/bench_prof.json 2> $R/$out/bench_prof.err
cd $R
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1); cp $f $out/bench_kernel_stats.csv 2>/dev/null
head -14 $out/bench_kernel_stats.csv | cut -c1-170
echo "== pmc" ; date
cd /tmp && PMC_WINDOWS=112 PMC_ALGO=$R/$out/pmc_algo.json timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_fetch -- python $R/tools/pmc_probe.py > $R/$out/pmc_fetch.log 2>&1
PMC_WINDOWS=112 timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_write -- python $R/tools/pmc_probe.py > $R/$out/pmc_write.log 2>&1
cd $R
python tools/pmc_summary.py /tmp/pmc_fetch /tmp/pmc_write $out/pmc_algo.json $out/r02_pmc.json
echo "== workloads" ; date
timeout 400 python bench.py --workload shard256 --steps 1 --warmup 1 > $out/shard256.json 2> $out/shard256.err; tail -c 600 $out/shard256.json
timeout 400 python bench.py --workload beam5 --steps 4 --warmup 1 > $out/beam5.json 2> $out/beam5.err; tail -c 600 $out/beam5.json
timeout 400 python bench.py --workload v3stream --steps 16 --warmup 1 --no-cpu-baseline > $out/v3.json 2> $out/v3.err; head -c 400 $out/v3.json
date
