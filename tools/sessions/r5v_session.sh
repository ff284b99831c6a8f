#!/bin/bash
# Round 5, session V (the round's last): the evidence of the final commit -- rocprofv3 statistics of the default command with ONE context in flight
# (at 448 windows the two contexts' kernels overlap inside a traced run too, so the two-in-flight statistics no longer give a kernel's own duration:
# profiles/r05_kernel_stats_two_in_flight.csv, session U), the driver's invocation, the default line, the GPU suite.
out=gpurun_out/${1:-r5v}; mkdir -p $out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T0=$(date +%s)
timeout 300 python -c "from whisper_amd import canary; canary.run_all()" 2>&1 | tail -1 | tee $out/canary.log
grep -q "mel ok" $out/canary.log || { echo "CANARY FAILED"; exit 3; }
echo "== rocprof of the default bench, one context in flight"; date
rm -rf /tmp/prof_bench1
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench1 -- python $R/bench.py --inflight 1 --no-roofline --no-cpu-baseline --no-single-stream --no-large --no-boundary --no-ids-check --no-small-job > $R/$out/bench_prof_one_context.json 2> $R/$out/bench_prof_one_context.err
cd $R
f=$(find /tmp/prof_bench1 -name "*kernel_stats.csv" | head -1); cp $f $out/bench_kernel_stats_one_context.csv 2>/dev/null
head -6 $out/bench_kernel_stats_one_context.csv | cut -c1-150
echo "== bench (driver invocation)"; date
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_k20.json 2> $out/bench_k20.err; echo "bench k20 rc=$?"; tail -2 $out/bench_k20.err
python -c "
import json; d=json.load(open('$out/bench_k20.json')); print({k:d[k] for k in ('value','ms_per_step','steps','warmup')}, d['config']['clips_per_step'], json.dumps(d['roofline']['end_to_end'])[:120]); print('small_job', d['small_job']['value'], 'large', (d.get('large_v2') or {}).get('value'), 'boundary', (d.get('through_boundary') or {}).get('value'), 'single', (d.get('single_stream') or {}).get('value'), 'ids', d['parity']['timed_ids']['consistent'])"
echo "== bench (default)"; date
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; tail -2 $out/bench.err
python -c "
import json; d=json.load(open('$out/bench.json')); print({k:d[k] for k in ('value','ms_per_step','steps','warmup')}, 'small_job', d['small_job']['value'], 'e2e', d['roofline']['end_to_end']['frac'], 'top', d['roofline']['kernel'], d['roofline']['frac'], 'hbm', d['roofline']['hbm_kernel']['frac'], d['roofline']['hbm_kernel']['avg_launch_us'])"
left=$(( ${2:-560} - ( $(date +%s) - T0 ) ))
echo "== tests ($left s left of the session's allowance)"; date
if [ $left -gt 60 ]; then
  timeout $left python -m pytest tests -m gpu -q -x > $out/test.log 2>&1; echo "pytest rc=$?" | tee -a $out/test.log
  tail -3 $out/test.log
fi
date
