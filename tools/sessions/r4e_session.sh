#!/bin/bash
# Round 4, session E: the vendor library's kernels on the encoder's shapes (names = tile configuration; effective clock; MFMA busy) next to gemmTiled8's counters
out=gpurun_out/r4E; mkdir -p $out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 300 python -c "from whisper_amd import canary; canary.run_all()" 2>&1 | tail -1
echo "== groupM (variant + 100 * groupM)"; PROBE_VARIANTS=40,240,640,840 PROBE_ROUNDS=3 PROBE_SHAPES=168000x4096x1024,168000x1024x1024,168000x1024x4096,168000x3072x1024 timeout 300 python tools/gemm8_probe.py 2>&1 | tee $out/groupm.txt
echo "== pmc gemmTiled8"; timeout 900 python tools/pmc_gemm.py run $out/pmc 2>&1 | grep -v "^columns" | tail -12
echo "== hipBLASLt kernel names + counters"
cd /tmp
for C in "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU SQ_INSTS_LDS"; do
  d=/tmp/hbl_$(echo $C | cut -c1-12 | tr ' ' _)
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $d -- python $R/tools/hipblaslt_ref.py > /dev/null 2>&1
  python - "$d" "$C" <<'PY'
import csv, glob, sys, os, collections
d, counters = sys.argv[1], sys.argv[2].split()
acc = collections.OrderedDict()
for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        n = row["Kernel_Name"]
        if "Cijk" not in n and "gemm" not in n.lower():
            continue
        key = (n[:220], row["Grid_Size"], row["Workgroup_Size"], row["LDS_Block_Size"], row["VGPR_Count"], row["Accum_VGPR_Count"])
        a = acc.setdefault(key, collections.defaultdict(float))
        a[row["Counter_Name"]] += float(row["Counter_Value"])
        a["n@" + row["Counter_Name"]] += 1
        a["ns@" + row["Counter_Name"]] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
for key, a in acc.items():
    print("KERNEL", key)
    for c in counters:
        if a.get("n@" + c):
            n = a["n@" + c]
            print("   %-32s per launch %.6g   avg duration %.1f us   launches %d" % (c, a[c] / n, a["ns@" + c] / n / 1e3, n))
PY
done 2>&1 | tee $R/$out/hipblaslt_counters.txt
cd $R
