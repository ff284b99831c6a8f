#!/bin/bash
# Round 4, session L: gemmTiled4 for single epilogues next to gemmTiled8 + lean epilogue (WH_GEMM_4WAVE_EPIS = bit mask over eEpilogue), bench A/B on one box
out=gpurun_out/${TAG:-r4L}; mkdir -p $out; export TMPDIR=/tmp
timeout 300 python -c "from whisper_amd import canary; canary.run_all()" 2>&1 | tail -1
run() { WH_GEMM_4WAVE_EPIS=$2 timeout 400 python bench.py --steps 32 --warmup 1 --no-cpu-baseline --no-single-stream --no-large --no-boundary > $out/$1.json 2> $out/$1.err
  python - <<PY
import json
d=json.load(open("$out/$1.json")); k=d["kernels"]["gemmTiled"]
print("%-10s %8.1f audio-s/s  %7.3f ms/step   gemmTiled %7.1f us per launch (%d launches)  mfma frac %s" % ("$1", d["value"], d["ms_per_step"], k["avg_us"], k["calls"], d["roofline"]["mfma_kernel"].get("frac")))
PY
}
for m in ${MASKS:-0 8 0 8 9 0}; do run m$m $m; done
