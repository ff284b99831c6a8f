#!/bin/bash
tag=${1:-s15}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TCP|TCC|TA|TD|GRBM)_[A-Z0-9_]+" | sort -u > $out/counters.txt
wc -l $out/counters.txt
grep -E "LDS|MFMA|WAIT|BUSY|STALL|ACTIVE" $out/counters.txt | tr '\n' ' ' | head -c 4000; echo
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_ACTIVE_INST_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$i -- python $R/tools/gemm_one.py 25 42000 4096 1024 > $out/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" "$set" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    if "gemmTiled" in r["Kernel_Name"] and "Lb1EEELb1" in r["Kernel_Name"] or ("gemmTiled" in r["Kernel_Name"] and "true>, true" in r["Kernel_Name"]):
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (s, n) in acc.items():
    print("  %-32s per launch %.4g  (%d launches)" % (k, s / max(n, 1), n))
if not acc:
    print("  no gemmTiled rows; kernels:", sorted({r["Kernel_Name"][:60] for r in rows})[:5])
PY
  else echo "set $i: no counter file"; tail -3 $out/pmc_$i.log; fi
done
