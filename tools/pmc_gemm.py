"""Counter passes over the production encoder product (gemmTiled8) on the model's shapes: settles what clock the chip runs this kernel at
and how busy the matrix pipe is.  bash: python tools/pmc_gemm.py run <out dir>   (on the GPU box; drives rocprofv3 itself, one pass per
counter group, never together with a trace domain other than --kernel-trace)
    effective clock  = GRBM_GUI_ACTIVE / kernel duration            (MI355X_MICROARCH.md, DVFS give-back)
    MFMA pipe busy   = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x CUs x GRBM_GUI_ACTIVE)   (cycles; 32 per v_mfma_f32_32x32x16_f16)
    MFMA ops         = SQ_INSTS_VALU_MFMA_MOPS_F16 x 512 FLOP, against 2 M N K
"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = [["GRBM_GUI_ACTIVE"], ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"], ["SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_INSTS_VALU", "SQ_INSTS_LDS"],
          ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"], ["FETCH_SIZE"], ["WRITE_SIZE"]]
SHAPES = "168000x4096x1024,168000x1024x1024,168000x1024x4096,168000x3072x1024"
# PMC_VARIANT=50 PMC_KERNEL=gemmTiled4: the same passes over the 4-wave kernel (profiles/r04_gemm4_probe.txt)
VARIANT = os.environ.get("PMC_VARIANT", "40")
KERNEL = os.environ.get("PMC_KERNEL", "gemmTiled8")


def run(out):
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, PROBE_VARIANTS=VARIANT, PROBE_ROUNDS="1", PROBE_SHAPES=SHAPES, TMPDIR="/tmp")
    # un-profiled timing of the same command first (a profiled pass clocks lower: never compare the two)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm8_probe.py")], env=env, stdout=subprocess.PIPE, text=True, cwd="/tmp")
    open(os.path.join(out, "unprofiled.txt"), "w").write(r.stdout)
    print(r.stdout)
    res = {}
    for i, g in enumerate(GROUPS):
        d = "/tmp/pmcg_%d" % i
        subprocess.run(["rocprofv3", "--pmc"] + g + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
                        os.path.join(ROOT, "tools", "gemm8_probe.py")], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd="/tmp")
        dur = {}
        for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for row in csv.DictReader(open(path)):
                if KERNEL in row["Kernel_Name"]:
                    dur[row["Dispatch_Id"]] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(path)):
                if KERNEL not in row["Kernel_Name"]:
                    continue
                # the probe runs the shapes in order, 10 timed + 1 warm-up launches each: key by grid/dispatch order is fragile, key by duration bucket instead
                ns = dur.get(row["Dispatch_Id"], 0)
                if not ns and row.get("End_Timestamp") and row.get("Start_Timestamp"):
                    ns = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
                k = res.setdefault("%d:%s" % (i, row["Dispatch_Id"]), {"ns": ns})
                k[row["Counter_Name"]] = k.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        for path in glob.glob(os.path.join(d, "**", "*.csv"), recursive=True)[:2]:
            print("columns of", os.path.basename(path), ":", open(path).readline().strip()[:400])
    json.dump(res, open(os.path.join(out, "raw.json"), "w"))
    summarize(out)


def summarize(out):
    raw = json.load(open(os.path.join(out, "raw.json")))
    # the probe runs the shapes in the order of PROBE_SHAPES, 2 warm-up + 10 timed launches each: dispatch order identifies the shape
    # (durations do not: a profiled pass jitters by +-20 %)
    shapes = []
    for sh in SHAPES.split(","):
        M, N, K = (int(x) for x in sh.split("x"))
        shapes.append(("N=%d K=%d" % (N, K), 2.0 * M * N * K))
    table = {s: {} for s, _ in shapes}
    by_pass = {}
    for k, v in raw.items():
        p, disp = k.split(":")
        by_pass.setdefault(p, []).append((int(disp), v))
    for p, rows in by_pass.items():
        rows = [v for _, v in sorted(rows, key=lambda kv: kv[0])]
        per = len(rows) // len(shapes)
        if per * len(shapes) != len(rows) or per < 3:
            print("pass", p, "has", len(rows), "launches of gemmTiled8: not", len(shapes), "equal groups")
            continue
        for i, (sname, flops) in enumerate(shapes):
            sel = rows[i * per + 2:(i + 1) * per]          # without the two warm-up launches
            t = table[sname]
            for c in sel[0]:
                if c == "ns":
                    continue
                t[c] = sum(r[c] for r in sel) / len(sel)
                t["ns@" + c] = sum(r["ns"] for r in sel) / len(sel)
            t["flops"] = flops
    lines = []
    for s, t in table.items():
        if "GRBM_GUI_ACTIVE" not in t:
            continue
        ns = t["ns@GRBM_GUI_ACTIVE"]
        clk = t["GRBM_GUI_ACTIVE"] / 8.0 / ns       # the counter is the sum over the 8 XCDs' GRBMs (the vendor kernel reads 1.86-1.97 GHz this way)
        line = "%-14s %8.1f us  %7.1f TFLOP/s (profiled pass)  effective clock %.3f GHz" % (s, ns / 1e3, t["flops"] / ns / 1e3, clk)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in t:
            ns2 = t["ns@SQ_VALU_MFMA_BUSY_CYCLES"]
            busy = t["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * 256 * clk * ns2)
            line += "  MFMA busy %.3f of SIMD-cycles (%.3g cycles; floor 2MNK/512*32 = %.3g)" % (busy, t["SQ_VALU_MFMA_BUSY_CYCLES"], t["flops"] / 512 / 2 * 32 / 16)
        if "SQ_INSTS_VALU_MFMA_MOPS_F16" in t:
            line += "  MOPS_F16 %.4g (x512 = %.4g FLOP vs %.4g)" % (t["SQ_INSTS_VALU_MFMA_MOPS_F16"], 512.0 * t["SQ_INSTS_VALU_MFMA_MOPS_F16"], t["flops"])
        for c in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "FETCH_SIZE", "WRITE_SIZE"):
            if c in t:
                line += "  %s %.4g" % (c, t[c])
        lines.append(line)
        print(line)
    open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
    json.dump(table, open(os.path.join(out, "table.json"), "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        summarize(sys.argv[2])
