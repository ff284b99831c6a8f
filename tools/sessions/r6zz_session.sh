#!/bin/bash
# Round 6, session ZZ (the FINAL build: + reorderCacheGroup, gemm_big_min_rows 8192 on top of session W's build: + gemmDecTile for the wide products of 33 .. 128 rows, the K-split MLP down-projection, attentionDecM, the vocabulary product on 64 x 64 tiles): the round's records on one box and one build (gemm_mf16, gemmDecTile, packed Q/K/V epilogue) -- GPU suite plain and under WH_DEBUG_POISON=0xFF, smoke(),
# the default bench line, the driver's invocation, kernel statistics of the bench command with one context in flight, PMC traffic at 448 windows, the two-rank dry run.
out=gpurun_out/${1:-r6zz}; mkdir -p $out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
date
timeout 1500 python -m pytest tests -m gpu -q -rP > $out/test.log 2>&1; echo "pytest rc=$?" | tee -a $out/test.log
grep -E "passed|failed|FAILED|^ERROR" $out/test.log | tail -8
WH_DEBUG_POISON=0xFF timeout 1500 python -m pytest tests -m gpu -q > $out/test_poison.log 2>&1; echo "pytest(poison) rc=$?" | tee -a $out/test_poison.log
grep -E "passed|failed|FAILED|^ERROR|WH_GUARD_VIOLATION" $out/test_poison.log | tail -8
timeout 300 python -c "import __graft_entry__ as e; e.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/smoke.log
date
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver.json 2> $out/bench_driver.err; echo "bench(driver) rc=$?"
python -c "
import json
for f in ('bench.json','bench_driver.json'):
    d=json.loads(open('$out/'+f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['value_r04_definition'], d['roofline']['frac'], d['roofline']['end_to_end']['frac'])"
date
rm -rf /tmp/prof_bench
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $R/bench.py --inflight 1 --steps 5 --warmup 1 --no-roofline --no-cpu-baseline --no-single-stream --no-large --no-boundary --no-workloads --no-small-job --no-ids-check > $R/$out/bench_prof.json 2> $R/$out/bench_prof.err
cd $R
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1); cp $f $out/bench_kernel_stats.csv 2>/dev/null
head -8 $out/bench_kernel_stats.csv | cut -c1-150
date
cd /tmp && PMC_WINDOWS=448 PMC_ALGO=$R/$out/pmc_algo.json timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_fetch -- python $R/tools/pmc_probe.py > $R/$out/pmc_fetch.log 2>&1
PMC_WINDOWS=448 timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_write -- python $R/tools/pmc_probe.py > $R/$out/pmc_write.log 2>&1
cd $R
python tools/pmc_summary.py /tmp/pmc_fetch /tmp/pmc_write $out/pmc_algo.json $out/r_pmc.json 2>&1 | tail -12
date
bash tools/dry_run_2ranks.sh ${1:-r6zz} 2>&1 | tail -6
date
rm -rf /tmp/prof_beam
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_beam -- python $R/bench.py --workload beam5 --model large-v2 --no-cpu-baseline --steps 2 --warmup 1 > $R/$out/beam_prof.json 2> $R/$out/beam_prof.err
cd $R
f=$(find /tmp/prof_beam -name "*kernel_stats.csv" | head -1); cp $f $out/beam_kernel_stats.csv 2>/dev/null
head -12 $out/beam_kernel_stats.csv | cut -c1-150
timeout 600 python tools/wide_time.py > $out/wide_time.txt 2>&1
timeout 600 python tools/deep_time.py > $out/deep_time.txt 2>&1
timeout 600 python tools/vocab_time.py > $out/vocab_time.txt 2>&1
timeout 600 python tools/cross_time.py > $out/cross_time.txt 2>&1
date
