#!/bin/bash
# Round 4, session H: the suite on the fixed beam decoder, the MT8 decode-product variant (tests + A/B), LDS counters of attentionEncT
out=gpurun_out/r4H; mkdir -p $out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 300 python -c "from whisper_amd import canary; canary.run_all()" 2>&1 | tail -1
echo "== tests"; timeout 1500 python -m pytest tests -m gpu -q -rP > $out/test.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED|Error" $out/test.log | tail -8; grep -E "beam 5 on" $out/test.log
DEF=$(python -c "from whisper_amd import binding as b; print(b.TUNE_DEFAULT)")
MT8=$(python -c "from whisper_amd import binding as b; print(b.TUNE_DEFAULT | b.TUNE_GEMV_MT8)")
run() { WH_TUNING=$2 timeout 400 python bench.py --steps 32 --warmup 1 --no-cpu-baseline --no-single-stream --no-large --no-boundary > $out/$1.json 2> $out/$1.err
  python - <<PY
import json
d=json.load(open("$out/$1.json")); k=d["kernels"]["gemvFused"]
print("%-10s %8.1f audio-s/s  %7.3f ms/step   gemvFused %6.2f us per launch (%d launches)" % ("$1", d["value"], d["ms_per_step"], k["avg_us"], k["calls"]))
PY
}
echo "== MT8 A/B"; run def1 $DEF; run mt8a $MT8; run def2 $DEF; run mt8b $MT8
echo "== LDS counters, encoder attention"
cd /tmp
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  d=/tmp/att_$(echo $C | cut -c1-14 | tr ' ' _)
  PMC_WINDOWS=112 PMC_STEPS=1 timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $d -- python $R/tools/pmc_probe.py > /dev/null 2>&1
  python - "$d" <<'PY'
import csv, glob, sys, os, collections
acc = collections.OrderedDict()
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        n = row["Kernel_Name"]
        if "attentionEnc" not in n: continue
        a = acc.setdefault(n[:60], collections.defaultdict(float))
        a[row["Counter_Name"]] += float(row["Counter_Value"]); a["n@"+row["Counter_Name"]] += 1
        a["ns@"+row["Counter_Name"]] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
for k, a in acc.items():
    print(k, {c: "%.4g per launch (%.0f us)" % (a[c]/a["n@"+c], a["ns@"+c]/a["n@"+c]/1e3) for c in a if "@" not in c})
PY
done 2>&1 | tee $R/$out/attention_counters.txt
