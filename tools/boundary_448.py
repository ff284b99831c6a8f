import sys, json
sys.path.insert(0, '.')
import bench
for (steps, C) in ((64, 32), (128, 64)):
    r = bench.through_boundary("medium", steps, C, 2, 7)
    print(steps, C, r["value"], r["seconds"], r["slots_per_group"], r["groups"], r["all_streams_ok"], flush=True)
