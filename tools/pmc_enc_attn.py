"""GPU: hardware counters of one encoder-attention kernel variant (ENC_MODE = enc_exp value), one rocprofv3 --pmc pass per group (never with another trace domain).
Usage (GPU box): python tools/pmc_enc_attn.py <out dir> [mode=2]"""
import csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = [["GRBM_GUI_ACTIVE", "SQ_WAVES"], ["SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES"], ["SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_INSTS_SALU"],
          ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY"], ["SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY"], ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT"],
          ["SQ_INST_CYCLES_VMEM", "SQ_INSTS_VMEM", "SQ_ACTIVE_INST_VMEM"], ["SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VALU_TRANS_F32" ]]


def main():
    out = sys.argv[1]
    mode = sys.argv[2] if len(sys.argv) > 2 else "2"
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, ENC_MODES=mode, TMPDIR="/tmp")
    avail = subprocess.run(["rocprofv3", "--list-avail"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd="/tmp").stdout
    open(os.path.join(out, "list_avail.txt"), "w").write(avail)
    res = {}
    for i, g in enumerate(GROUPS):
        g = [c for c in g if c in avail]
        if not g:
            continue
        d = "/tmp/pmce_%d" % i
        subprocess.run(["rocprofv3", "--pmc"] + g + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "tools", "enc_attn_time.py")],
                       env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd="/tmp")
        rows = {}
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(path)):
                if "attentionEnc" not in row["Kernel_Name"]:
                    continue
                rows.setdefault(row["Dispatch_Id"], {})
                rows[row["Dispatch_Id"]][row["Counter_Name"]] = rows[row["Dispatch_Id"]].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        n = len(rows)
        for c in g:
            vals = [r.get(c, 0.0) for r in rows.values()]
            if vals:
                res[c] = sum(vals) / len(vals)
        print("pass", i, g, "launches", n, {c: res.get(c) for c in g}, flush=True)
    json.dump(res, open(os.path.join(out, "enc_attn_pmc_mode%s.json" % mode), "w"), indent=1)
    act = res.get("GRBM_GUI_ACTIVE", 0)
    if act:
        print("per launch: GRBM_GUI_ACTIVE %.3g cycles; MFMA busy %.3f of 4 x 256 SIMD-cycles; VALU insts per wave-cycle ..." % (act, res.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * 256 * act)))


if __name__ == "__main__":
    main()
