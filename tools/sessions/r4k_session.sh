#!/bin/bash
# Round 4, session K: gemmTiled4 (one wave per SIMD) -- probe variants against gemmTiled8, op tests, model tests and bench A/B under the new tuning bit
out=gpurun_out/${TAG:-r4K}; mkdir -p $out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 300 python -c "from whisper_amd import canary; canary.run_all()" 2>&1 | tail -1
echo "== probe"; date
PROBE_VARIANTS=${PROBE_VARIANTS:-40,50,51,52,53,54} PROBE_ROUNDS=${PROBE_ROUNDS:-3} PROBE_SHAPES=${PROBE_SHAPES:-168000x1024x1024,168000x3072x1024,168000x4096x1024,168000x1024x4096,16397x1000x192,42000x1280x1280} \
  timeout 400 python tools/gemm8_probe.py 2>&1 | tee $out/probe.txt
grep -q "^GEMM" $out/probe.txt || { echo "PROBE FAILED"; exit 3; }
echo "== op tests"; date
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "4wave or big_tiles" 2>&1 | tail -5
[ -z "$FULL" ] && exit 0
DEF=$(python -c "from whisper_amd import binding as b; print(b.TUNE_DEFAULT & ~(b.TUNE_GEMM_4WAVE | b.TUNE_GEMM_FAST_EPI))")
W4=$(python -c "from whisper_amd import binding as b; print((b.TUNE_DEFAULT & ~b.TUNE_GEMM_FAST_EPI) | b.TUNE_GEMM_4WAVE)")
FE=$(python -c "from whisper_amd import binding as b; print((b.TUNE_DEFAULT & ~b.TUNE_GEMM_4WAVE) | b.TUNE_GEMM_FAST_EPI)")
echo "== model-level tests with ${MODEL_TUNING:-W4}"; date
MT=$W4; [ "$MODEL_TUNING" = FE ] && MT=$FE
WH_TUNING=$MT timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -4
run() { WH_TUNING=$2 timeout 400 python bench.py --steps 32 --warmup 1 --no-cpu-baseline --no-single-stream --no-large --no-boundary > $out/$1.json 2> $out/$1.err
  python - <<PY
import json
d=json.load(open("$out/$1.json")); k=d["kernels"]["gemmTiled"]
print("%-10s %8.1f audio-s/s  %7.3f ms/step   gemmTiled %7.1f us per launch (%d launches)  mfma frac %s" % ("$1", d["value"], d["ms_per_step"], k["avg_us"], k["calls"], d["roofline"]["mfma_kernel"].get("frac")))
PY
}
echo "== bench A/B"; date
if [ "$MODEL_TUNING" = FE ]; then run def1 $DEF; run fea $FE; run def2 $DEF; run feb $FE; else run def1 $DEF; run w4a $W4; run def2 $DEF; run w4b $W4; fi
date
