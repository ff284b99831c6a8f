#!/bin/bash
# Round 5, session F: the round's evidence -- full GPU suite, default bench line, rocprofv3 statistics of the same command, PMC passes at the timed batch
# size, the driver's invocation, the other BASELINE workloads, smoke.
out=gpurun_out/${1:-r5f}; mkdir -p $out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 python -c "from whisper_amd import canary; canary.run_all()" 2>&1 | tee $out/canary.log
grep -q "mel ok" $out/canary.log || { echo "CANARY FAILED"; exit 3; }
echo "== tests"; date
timeout 1500 python -m pytest tests -m gpu -q -rP > $out/test.log 2>&1; echo "pytest rc=$?" | tee -a $out/test.log
grep -E "passed|failed|FAILED|Error" $out/test.log | tail -12
grep -E "ranks . on . device|PARITY MODE|lock step vs" $out/test.log | tail -12
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as e; e.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/smoke.log
echo "== bench (default)"; date
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; tail -3 $out/bench.err
python - <<PY
import json
d=json.load(open("$out/bench.json"))
print({k:d[k] for k in ("value","ms_per_step","steps")}, d["config"]["batch_plan"])
r=d["roofline"]
print("top", {k:r[k] for k in ("kernel","bound","achieved","frac","traffic")})
for k in ("mfma_kernel","hbm_kernel","encoder_attention","decode_chain","end_to_end"): print(k, json.dumps(r.get(k))[:330])
print("parity.timed_ids", json.dumps(d["parity"].get("timed_ids"))[:400])
for k in ("through_boundary","single_stream"): print(k, json.dumps(d.get(k))[:200])
l=d.get("large_v2") or {}
print("large_v2", {k:l.get(k) for k in ("value","ms_per_step","batch_plan","error")}, json.dumps(l.get("beam5"))[:200])
PY
echo "== rocprof of the default bench"; date
rm -rf /tmp/prof_bench
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $R/bench.py --no-roofline --no-cpu-baseline --no-single-stream --no-large --no-boundary --no-ids-check > $R/$out/bench_prof.json 2> $R/$out/bench_prof.err
cd $R
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1); cp $f $out/bench_kernel_stats.csv 2>/dev/null
head -16 $out/bench_kernel_stats.csv | cut -c1-170
echo "== pmc at 224 windows"; date
cd /tmp && PMC_WINDOWS=224 PMC_ALGO=$R/$out/pmc_algo.json timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_fetch -- python $R/tools/pmc_probe.py > $R/$out/pmc_fetch.log 2>&1
PMC_WINDOWS=224 timeout 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_write -- python $R/tools/pmc_probe.py > $R/$out/pmc_write.log 2>&1
cd $R
python tools/pmc_summary.py /tmp/pmc_fetch /tmp/pmc_write $out/pmc_algo.json $out/r_pmc.json 2>&1 | tail -14
echo "== bench (driver invocation)"; date
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_k20.json 2> $out/bench_k20.err; echo "bench k20 rc=$?"
python -c "
import json; d=json.load(open('$out/bench_k20.json')); print({k:d[k] for k in ('value','ms_per_step','steps')}, d['config']['batch_plan'], json.dumps(d['roofline']['end_to_end'])[:200]); print('large', (d.get('large_v2') or {}).get('value'), 'boundary', (d.get('through_boundary') or {}).get('value'))"
echo "== workloads"; date
timeout 400 python bench.py --workload shard256 --steps 1 --warmup 1 --batch 128 > $out/shard256.json 2> $out/shard256.err; tail -c 400 $out/shard256.json; echo
timeout 400 python bench.py --workload beam5 --steps 8 --warmup 2 > $out/beam5.json 2> $out/beam5.err; tail -c 300 $out/beam5.json; echo
timeout 400 python bench.py --workload v3stream --no-cpu-baseline --no-single-stream --no-boundary --no-large > $out/v3.json 2> $out/v3.err; head -c 300 $out/v3.json; echo
date
