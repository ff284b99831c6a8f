#!/bin/bash
# Round 4, session I: the whole suite (spread sampler, table attention threshold, batch stress, MT8 op test), then single-stream A/B of the spread sampler
out=gpurun_out/r4I; mkdir -p $out; export TMPDIR=/tmp
timeout 300 python -c "from whisper_amd import canary; canary.run_all()" 2>&1 | tail -1
echo "== tests"; timeout 1800 python -m pytest tests -m gpu -q -rP > $out/test.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED|Error" $out/test.log | tail -8
DEF=$(python -c "from whisper_amd import binding as b; print(b.TUNE_DEFAULT)")
ONE=$(python -c "from whisper_amd import binding as b; print(b.TUNE_DEFAULT & ~b.TUNE_SAMPLE_SPREAD)")
for t in $ONE $DEF $ONE $DEF; do echo "tuning $t"; SS_MODEL_FILE=/tmp/ss_model.bin WH_TUNING=$t timeout 300 python tools/single_stream_prof.py 2>&1 | grep run_full; done | tee $out/single_stream_ab.txt
