"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/pmc_probe.py -> profiles/r02_pmc.json, keyed by the kernel classes
bench.py reports (its `roofline.traffic` reads this file):

    python tools/pmc_summary.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <PMC_ALGO json> [out json]

Counter unit = KB. HBM read bytes = 2 x FETCH_SIZE x 1024 (gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes for wide
coalesced reads, MI355X_MICROARCH.md 'HBM'); WRITE_SIZE x 1024 as reported (uncalibrated there)."""
import csv
import glob
import json
import os
import sys

CLASSES = [("attentionDecCross", lambda n: "attentionDecG" in n and "Lb1" in n or ("attentionDecG<" in n and ", true>" in n)),
           ("attentionDec", lambda n: "attentionDec" in n or "selfAttnDecWave" in n), ("selfBlockDec", lambda n: "selfBlockDec" in n),
           ("gemvFused", lambda n: "gemvFused" in n or "gemmDecRows" in n or "gemmAllRows" in n), ("gemmTiled", lambda n: "gemmTiled" in n), ("gemmSkinny", lambda n: "gemmSkinny" in n),
           ("attentionEnc", lambda n: "attentionEnc" in n), ("layerNorm", lambda n: "layerNormKernel" in n), ("mel", lambda n: "melKernel" in n),
           ("softMaxSample", lambda n: "softMaxSample" in n), ("vocabSoftMax", lambda n: "softMaxRows" in n), ("embed", lambda n: "embedKernel" in n)]


def classify(name):
    for c, f in CLASSES:
        if f(name):
            return c
    return None


def read(dirname, counter):
    acc = {}
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                c = classify(row["Kernel_Name"])
                if c is None:
                    continue
                a = acc.setdefault(c, [0, 0.0, {}])
                a[0] += 1
                a[1] += float(row["Counter_Value"])
                k = row["Kernel_Name"][:120]
                a[2][k] = a[2].get(k, 0) + 1
    return acc


def main():
    fetch_dir, write_dir, algo_path = sys.argv[1:4]
    out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02_pmc.json")
    fetch, write = read(fetch_dir, "FETCH_SIZE"), read(write_dir, "WRITE_SIZE")
    algo = json.load(open(algo_path))
    # kernel names do not tell the encoder's tiles / LayerNorm launches from the decoder graph's (classes gemmDecode / layerNormDec of the library's own
    # accounting, round 5): the name classes hold both
    for name_cls, extra in (("gemmTiled", "gemmDecode"), ("layerNorm", "layerNormDec")):
        a, b = algo["classes"].get(name_cls), algo["classes"].get(extra)
        if a and b:
            for f in ("calls", "ms", "flops", "bytes"):
                a[f] += b[f]
    kernels = {}
    for c, (n, kb, names) in fetch.items():
        wn, wkb, _ = write.get(c, (0, 0.0, {}))
        a = algo["classes"].get(c)
        e = {"launches": n, "hbm_read_bytes_per_launch": round(2.0 * kb * 1024 / n), "hbm_write_bytes_per_launch": round(wkb * 1024 / wn) if wn else 0,
             "kernel_names": names}
        if a and a["calls"]:
            e["algorithmic_bytes_per_launch"] = round(a["bytes"] / a["calls"])
            e["algorithmic_flops_per_launch"] = round(a["flops"] / a["calls"])
            e["launches_algorithmic"] = a["calls"]
            e["traffic_over_algorithmic"] = round((e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]) / max(e["algorithmic_bytes_per_launch"], 1), 3)
        kernels[c] = e
    # kernel names cannot tell the prompt step's cross-attention (attentionDecG<.., false>, query projected by a separate product)
    # from its causal self-attention: the name class "attentionDec" holds both, so its algorithmic bytes are the library's
    # self-attention class plus the cross-attention launches that are not in the fused-query name class
    cross, self_, named = algo["classes"].get("attentionDecCross"), algo["classes"].get("attentionDec"), kernels.get("attentionDec")
    if cross and self_ and named and cross["calls"]:
        unfused = max(cross["calls"] - kernels.get("attentionDecCross", {}).get("launches", 0), 0)
        n = self_["calls"] + unfused
        if n:
            named["algorithmic_bytes_per_launch"] = round((self_["bytes"] + unfused * cross["bytes"] / cross["calls"]) / n)
            named["algorithmic_flops_per_launch"] = round((self_["flops"] + unfused * cross["flops"] / cross["calls"]) / n)
            named["launches_algorithmic"] = n
            named["traffic_over_algorithmic"] = round((named["hbm_read_bytes_per_launch"] + named["hbm_write_bytes_per_launch"]) / max(named["algorithmic_bytes_per_launch"], 1), 3)
            named["holds"] = "%d causal self-attention + %d unfused cross-attention launches of the prompt step" % (self_["calls"], unfused)
    doc = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on tools/pmc_probe.py: eager launches, %s shape, %d windows in lock "
                   "step, encoder + prompt step + %d single-token steps; read = 2 x FETCH_SIZE (gfx950 correction), write = WRITE_SIZE as reported; "
                   "per-launch averages over all launches of the class in that run" % (algo["model"], algo["windows"], algo["steps"]),
           "kernels": kernels}
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    for c, e in kernels.items():
        print("%-18s launches %5d  read %12d  write %12d  algorithmic %12s  x%s" % (c, e["launches"], e["hbm_read_bytes_per_launch"], e["hbm_write_bytes_per_launch"],
                                                                                  e.get("algorithmic_bytes_per_launch"), e.get("traffic_over_algorithmic")))


if __name__ == "__main__":
    main()
