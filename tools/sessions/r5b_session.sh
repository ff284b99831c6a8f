#!/bin/bash
# Round 5, session B: full GPU suite, the default bench line (64 passes), the driver's invocation (20 passes), kernel trace of the latter for the overlap analysis.
out=gpurun_out/${1:-r5b}; mkdir -p $out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 python -c "from whisper_amd import canary; canary.run_all()" 2>&1 | tee $out/canary.log
grep -q "mel ok" $out/canary.log || { echo "CANARY FAILED"; exit 3; }
echo "== tests"; date
timeout 1500 python -m pytest tests -m gpu -q -rP > $out/test.log 2>&1; echo "pytest rc=$?" | tee -a $out/test.log
grep -E "passed|failed|FAILED|Error|device-ranked" $out/test.log | tail -25
echo "== bench (default)"; date
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; tail -5 $out/bench.err
python - <<PY
import json
d=json.load(open("$out/bench.json"))
print({k:d[k] for k in ("value","ms_per_step","steps")}, d["config"]["batch_plan"])
r=d["roofline"]
print("top", {k:r[k] for k in ("kernel","bound","achieved","frac","traffic")})
for k in ("mfma_kernel","hbm_kernel","encoder_attention","decode_chain","end_to_end"): print(k, json.dumps(r.get(k))[:420])
print("parity.timed_ids", json.dumps(d["parity"].get("timed_ids"))[:600])
for k in ("through_boundary","single_stream"): print(k, json.dumps(d.get(k))[:300])
l=d.get("large_v2") or {}
print("large_v2", {k:l.get(k) for k in ("value","ms_per_step","batch_plan","error")}, json.dumps(l.get("beam5"))[:300])
print("large roofline", json.dumps({k:(l.get("roofline") or {}).get(k) for k in ("kernel","frac","decode_chain","end_to_end")})[:900])
PY
echo "== bench (driver invocation)"; date
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-large --no-single-stream --no-boundary --no-cpu-baseline > $out/bench_k20.json 2> $out/bench_k20.err; echo "bench k20 rc=$?"
python -c "
import json; d=json.load(open('$out/bench_k20.json')); print({k:d[k] for k in ('value','ms_per_step','steps')}, d['config']['batch_plan'], json.dumps(d['roofline']['end_to_end'])[:300])"
echo "== kernel trace of the driver invocation"; date
rm -rf /tmp/trace_k20
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_k20 -- python $R/bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline --no-single-stream --no-large --no-boundary --no-ids-check > $R/$out/bench_trace.json 2> $R/$out/bench_trace.err
cd $R
f=$(find /tmp/trace_k20 -name "*kernel_stats.csv" | head -1); cp $f $out/k20_kernel_stats.csv 2>/dev/null
head -12 $out/k20_kernel_stats.csv | cut -c1-170
t=$(find /tmp/trace_k20 -name "*kernel_trace.csv" | head -1); head -2 $t | cut -c1-400
python tools/overlap_trace.py /tmp/trace_k20 20 $out/k20_overlap.json 2>&1 | tail -60
date
