#!/bin/bash
# One GPU session on a gpurun box:  bash tools/gpu_session.sh <tag> [stage ...]
# Stages (default: canary tests smoke bench):
#   canary    torch-only op, then the library's smallest entry points (BAD_BOX if the lease itself is broken)
#   tests     pytest -m gpu (full suite, no -x)              -> $out/test.log
#   poison    the full suite again under WH_DEBUG_POISON=0xFF (stale-memory / guard-region check) -> $out/test_poison.log
#   smoke     __graft_entry__.smoke() in three fresh processes
#   bench     python bench.py                                 -> $out/bench.json
#   prof      rocprofv3 --kernel-trace --stats over bench.py  -> $out/bench_kernel_stats.csv
#   pmc       FETCH_SIZE / WRITE_SIZE passes (separate runs)  -> $out/r_pmc.json
#   workloads shard256 / beam5 / v3stream bench lines
#   extra     whatever $EXTRA_CMD holds (one-off probes)
tag=${1:-s}; shift
stages=${@:-canary tests smoke bench}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
has() { [[ " $stages " == *" $1 "* ]]; }

timeout 300 python -c "from whisper_amd import canary; canary.run_all()" 2>&1 | tee $out/canary.log
grep -q "mel ok" $out/canary.log || { echo "CANARY FAILED (see above: BAD_BOX = the lease, otherwise ours)"; exit 3; }

if has tests; then
  echo "== tests"; date
  timeout 1500 python -m pytest tests -m gpu -q -rP > $out/test.log 2>&1; echo "pytest rc=$?" | tee -a $out/test.log
  grep -E "passed|failed|FAILED|Error" $out/test.log | tail -15
fi
if has poison; then
  echo "== tests under WH_DEBUG_POISON=0xFF"; date
  WH_DEBUG_POISON=0xFF timeout 1500 python -m pytest tests -m gpu -q -rP > $out/test_poison.log 2>&1; echo "pytest(poison) rc=$?" | tee -a $out/test_poison.log
  grep -E "passed|failed|FAILED|Error|WH_GUARD_VIOLATION" $out/test_poison.log | tail -25
fi
if has smoke; then
  for i in 1 2 3; do
    timeout 300 python -c "import __graft_entry__ as e; e.smoke()" > $out/smoke$i.log 2>&1; echo "smoke$i rc=$?"; tail -1 $out/smoke$i.log
  done
fi
if has bench; then
  echo "== bench"; date
  timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
  python - <<PY
import json
d=json.load(open("$out/bench.json"))
print({k:d[k] for k in ("value","ms_per_step","steps")})
print(json.dumps(d["roofline"])[:2200])
for k in ("through_boundary","single_stream","large_v2","parity"): print(k, d.get(k))
PY
fi
if has prof; then
  echo "== rocprof bench"; date
  rm -rf /tmp/prof_bench
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $R/bench.py --no-roofline --no-cpu-baseline --no-single-stream --no-large --no-boundary > $R/$out/bench_prof.json 2> $R/$out/bench_prof.err
  cd $R
  f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1); cp $f $out/bench_kernel_stats.csv 2>/dev/null
  head -14 $out/bench_kernel_stats.csv | cut -c1-170
fi
if has pmc; then
  echo "== pmc"; date
  cd /tmp && PMC_WINDOWS=112 PMC_ALGO=$R/$out/pmc_algo.json timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_fetch -- python $R/tools/pmc_probe.py > $R/$out/pmc_fetch.log 2>&1
  PMC_WINDOWS=112 timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_write -- python $R/tools/pmc_probe.py > $R/$out/pmc_write.log 2>&1
  cd $R
  python tools/pmc_summary.py /tmp/pmc_fetch /tmp/pmc_write $out/pmc_algo.json $out/r_pmc.json
fi
if has workloads; then
  echo "== workloads"; date
  timeout 400 python bench.py --workload shard256 --steps 1 --warmup 1 > $out/shard256.json 2> $out/shard256.err; tail -c 600 $out/shard256.json
  timeout 400 python bench.py --workload beam5 --steps 8 --warmup 2 > $out/beam5.json 2> $out/beam5.err; tail -c 600 $out/beam5.json
  timeout 400 python bench.py --workload v3stream --steps 16 --warmup 1 --no-cpu-baseline > $out/v3.json 2> $out/v3.err; head -c 400 $out/v3.json
fi
if has extra; then
  echo "== extra: $EXTRA_CMD"; date
  bash -c "$EXTRA_CMD" > $out/extra.log 2>&1; echo "extra rc=$?"; tail -40 $out/extra.log
fi
date
