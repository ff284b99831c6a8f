#!/bin/bash
# Round 4: encoders of the contexts in flight one at a time (TUNE_ENC_SERIAL = 134217728) x contexts in flight x batch plan x CUs the
# persistent encoder product leaves to the neighbours' decode launches (WH_GEMM_SPARE_CUS).  bash tools/phase_probe.sh <tag>
tag=${1:-phase}; out=gpurun_out/$tag; mkdir -p $out
DEF=$(python -c "from whisper_amd import binding as b; print(b.TUNE_DEFAULT)")
SER=$(python -c "from whisper_amd import binding as b; print(b.TUNE_DEFAULT | b.TUNE_ENC_SERIAL)")
run() {  # name tuning spare steps inflight plan
  WH_TUNING=$2 WH_GEMM_SPARE_CUS=$3 timeout 300 python bench.py --steps $4 --warmup 1 --inflight $5 ${6:+--plan $6} --no-roofline --no-cpu-baseline --no-single-stream --no-large --no-boundary > $out/$1.json 2> $out/$1.err
  python - <<PY
import json
try:
    d=json.load(open("$out/$1.json")); print("%-34s %8.1f audio-s/s  %7.3f ms/step  plan %s" % ("$1", d["value"], d["ms_per_step"], d["config"]["batch_plan"]))
except Exception as e: print("$1 FAILED", e)
PY
}
run k20_default        $DEF 32 20 2
run k20_serial_s32     $SER 32 20 2
run k20_serial_s64     $SER 64 20 2
run k20_i4_noserial    $DEF 32 20 4 5,5,5,5
run k20_i4_serial_s32  $SER 32 20 4 5,5,5,5
run k20_i4_serial_s64  $SER 64 20 4 5,5,5,5
run k20_i3_serial_s32  $SER 32 20 3 7,7,6
run k20_i3_serial_s64  $SER 64 20 3 7,7,6
run k32_default        $DEF 32 32 2
run k32_serial_s32     $SER 32 32 2
run k32_i4_serial_s32  $SER 32 32 4 8,8,8,8
run k32_i4_serial_s64  $SER 64 32 4 8,8,8,8
run k128_default       $DEF 32 128 2
run k128_serial_s32    $SER 32 128 2
run k128_serial_s64    $SER 64 128 2
run k128_serial_s96    $SER 96 128 2
run k128_i3_serial_s64 $SER 64 128 3
run k128_i4_serial_s64 $SER 64 128 4 8,8,8,8,8,8,8,8,8,8,8,8,8,8,8,8
run k128_default_again $DEF 32 128 2
