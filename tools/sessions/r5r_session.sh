#!/bin/bash
# Round 5, session R: the bench line with a step = one lock-step batch: the driver's invocation, the default line, rocprofv3 statistics of the default command.
out=gpurun_out/${1:-r5r}; mkdir -p $out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 python -c "from whisper_amd import canary; canary.run_all()" 2>&1 | tail -1 | tee $out/canary.log
grep -q "mel ok" $out/canary.log || { echo "CANARY FAILED"; exit 3; }
echo "== bench (driver invocation)"; date
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_k20.json 2> $out/bench_k20.err; echo "bench k20 rc=$?"; tail -4 $out/bench_k20.err
python -c "
import json; d=json.load(open('$out/bench_k20.json')); print({k:d[k] for k in ('value','ms_per_step','steps','warmup')}, d['config']['clips_per_step'], d['config']['clip_passes'], json.dumps(d['roofline']['end_to_end'])[:200]); print('small_job', d['small_job']['value'], 'large', (d.get('large_v2') or {}).get('value'), 'boundary', (d.get('through_boundary') or {}).get('value'), 'single', (d.get('single_stream') or {}).get('value'), 'ids', d['parity']['timed_ids']['consistent'])"
echo "== bench (default)"; date
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; tail -3 $out/bench.err
python -c "
import json; d=json.load(open('$out/bench.json')); print({k:d[k] for k in ('value','ms_per_step','steps','warmup')}, d['config']['batch_plan'], 'small_job', d['small_job']['value'], 'e2e', d['roofline']['end_to_end']['frac'], 'top', d['roofline']['kernel'], d['roofline']['frac'])"
echo "== rocprof of the default bench"; date
rm -rf /tmp/prof_bench
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $R/bench.py --no-roofline --no-cpu-baseline --no-single-stream --no-large --no-boundary --no-ids-check --no-small-job > $R/$out/bench_prof.json 2> $R/$out/bench_prof.err
cd $R
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1); cp $f $out/bench_kernel_stats.csv 2>/dev/null
head -6 $out/bench_kernel_stats.csv | cut -c1-150
date
