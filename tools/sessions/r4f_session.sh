#!/bin/bash
# Round 4, session F: attentionEncT (exponential as a lookup in the reference's table, in LDS) -- op tests, then A/B in the bench
out=gpurun_out/r4F; mkdir -p $out; export TMPDIR=/tmp
timeout 300 python -c "from whisper_amd import canary; canary.run_all()" 2>&1 | tail -1
echo "== op tests"; timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "flash or exp_table" -rP 2>&1 | grep -E "table kernel|passed|failed|Error|flash_attention table" | tail -30
DEF=$(python -c "from whisper_amd import binding as b; print(b.TUNE_DEFAULT & ~b.TUNE_ATTN_ENC_TABLE)")
TAB=$(python -c "from whisper_amd import binding as b; print(b.TUNE_DEFAULT | b.TUNE_ATTN_ENC_TABLE)")
run() { WH_TUNING=$2 timeout 400 python bench.py --steps 32 --warmup 1 --no-cpu-baseline --no-single-stream --no-large --no-boundary > $out/$1.json 2> $out/$1.err
  python - <<PY
import json
d=json.load(open("$out/$1.json")); k=d["kernels"]["attentionEnc"]
print("%-10s %8.1f audio-s/s  %7.3f ms/step   attentionEnc %8.1f us per launch, %6.1f TFLOP/s" % ("$1", d["value"], d["ms_per_step"], k["avg_us"], k["tflops"]))
PY
}
run def1 $DEF; run tab1 $TAB; run def2 $DEF; run tab2 $TAB
echo "== model-level tests with the table kernel"; WH_TUNING=$TAB timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -3
