#!/bin/bash
# Round 5, session G: is the 2.18 x fabric traffic of the encoder product HBM or Infinity Cache? Memory-side request counters of the
# production gemmTiled8 on the MLP-up shape, and of a plain streaming kernel (LayerNorm over 1 GB) as the yardstick whose traffic is certainly DRAM.
out=gpurun_out/${1:-r5g}; mkdir -p $out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o -i -E "\b(TCC_[A-Z0-9_]*(DRAM|EA0_RDREQ|EA0_WRREQ|HIT|MISS|REQ|BUBBLE|MALL)[A-Za-z0-9_]*|[A-Z_]*MALL[A-Za-z0-9_]*)\b" | sort -u > $R/$out/counters_available.txt
wc -l $R/$out/counters_available.txt; head -80 $R/$out/counters_available.txt | tr '\n' ' '; echo
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_DRAM_sum" "TCC_EA0_WRREQ_DRAM_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum" "TCC_EA0_RDREQ_IO_sum" "TCC_EA0_RDREQ_GMI_sum" "FETCH_SIZE"; do
  d=/tmp/pmcx_$(echo $grp | tr ' ' '_')
  rm -rf $d
  PROBE_VARIANTS=40 PROBE_ROUNDS=1 PROBE_SHAPES=168000x4096x1024,168000x1024x4096 timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python $R/tools/gemm8_probe.py > /dev/null 2> $R/$out/err_$(echo $grp | tr ' ' '_').log
  python - "$d" "$grp" <<'PY'
import csv, glob, os, sys
d, grp = sys.argv[1], sys.argv[2]
acc = {}
for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        k = "gemmTiled8" if "gemmTiled8" in r["Kernel_Name"] else ("layerNorm" if "layerNorm" in r["Kernel_Name"] else None)
        if not k: continue
        a = acc.setdefault((k, r["Counter_Name"]), [])
        a.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
if not acc: print("%-44s no data (counter not available on this device?)" % grp)
for (k, c), rows in sorted(acc.items()):
    rows.sort()
    vals = [v for _, v in rows]
    half = len(vals) // 2
    print("%-12s %-28s launches %3d   first shape (N=4096,K=1024) mean %.6g   second shape (N=1024,K=4096) mean %.6g" % (k, c, len(vals), sum(vals[:half]) / max(half, 1), sum(vals[half:]) / max(len(vals) - half, 1)))
PY
done 2>&1 | tee $R/$out/summary.txt
cd $R
date
