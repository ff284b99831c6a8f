"""How much of the timed region's kernel time actually overlaps: reads a rocprofv3 --kernel-trace CSV of bench.py (columns Kernel_Name,
Start_Timestamp, End_Timestamp, Queue_Id ... as rocprofv3 writes them) and prints, for the LAST `passes x 7` spectrogram launches onwards
(= the timed region: every clip pass starts with 7 mel launches; nothing runs after it with --no-roofline --no-cpu-baseline ...):

    wall time, sum of kernel durations, union busy time, time with >= 2 kernels resident, idle time;
    per kernel class: own time, and how much of it ran while a kernel of ANOTHER queue was resident (by class of that other kernel).

    python tools/overlap_trace.py <dir with *_kernel_trace.csv> <clip passes in the timed region> [out.json]
"""
import csv
import glob
import json
import os
import sys

CLASSES = (("gemmTiled", "gemm"), ("attentionEncT", "attnEnc"), ("attentionEncF", "attnEnc"), ("attentionDecG<1, true", "crossAttn"), ("attentionDecG", "attnDec"),
           ("selfBlockDec", "selfBlock"), ("gemvFused", "decProducts"), ("gemmDecRows", "decProducts"), ("gemmAllRows", "vocab"), ("layerNorm", "layerNorm"),
           ("melKernel", "mel"), ("melNormalize", "mel"), ("softMaxSample", "sampler"), ("embed", "embed"))


def cls(name):
    for key, c in CLASSES:
        if key in name:
            return c
    return "other"


def main():
    d, passes = sys.argv[1], int(sys.argv[2])
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
    rows.sort()
    mel = [i for i, r in enumerate(rows) if "melKernel" in r[2]]
    first = mel[-passes * 7]
    rows = rows[first:]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    # sweep
    ev = []
    for i, (a, b, n, q) in enumerate(rows):
        ev.append((a, 1, i))
        ev.append((b, 0, i))
    ev.sort()
    active = set()
    last = t0
    busy = multi = 0
    own = {}
    over = {}
    for t, kind, i in ev:
        dt = t - last
        if dt > 0 and active:
            busy += dt
            if len(active) > 1:
                multi += dt
            for k in active:
                c = cls(rows[k][2])
                own[c] = own.get(c, 0) + dt
                others = {cls(rows[j][2]) for j in active if j != k and rows[j][3] != rows[k][3]}
                for o in others:
                    over.setdefault(c, {})[o] = over.setdefault(c, {}).get(o, 0) + dt
        last = t
        if kind:
            active.add(i)
        else:
            active.discard(i)
    total = sum(b - a for a, b, _, _ in rows)
    out = {"kernels": len(rows), "queues": sorted({r[3] for r in rows}), "wall_ms": (t1 - t0) / 1e6, "sum_of_kernel_ms": total / 1e6, "busy_ms": busy / 1e6,
           "two_or_more_resident_ms": multi / 1e6, "idle_ms": (t1 - t0 - busy) / 1e6,
           "per_class_ms": {c: round(v / 1e6, 2) for c, v in sorted(own.items(), key=lambda kv: -kv[1])},
           "overlapped_with_other_queue_ms": {c: {o: round(v / 1e6, 2) for o, v in sorted(m.items(), key=lambda kv: -kv[1])} for c, m in over.items()}}
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
