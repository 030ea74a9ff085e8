"""Summarise one tools/profile_bench.sh output directory: per-kernel time from the kernel trace statistics and
HBM-side bytes per launch from the FETCH_SIZE / WRITE_SIZE passes (both counters are in kilobytes).

FETCH_SIZE correction: gfx950 tallies every read request at 64 bytes.  Streaming kernels issue 128-byte requests, so their
figure must be doubled; the traversal and shading kernels read random 64-byte sectors (nodes, triangles, corners), for
which the raw figure is exact — tools/micro/calibrate_fetch.sh measures 1.02 x the useful bytes on random 64-byte record
gathers.  The table prints both; the json keeps the raw figure for the gather kernels."""
import collections
import os
import csv
import glob
import re
import sys

import json
root = sys.argv[1]
config = sys.argv[2] if len(sys.argv) > 2 else "readme_pbr"
bench = json.loads(open(root + "/bench.json").read().strip().splitlines()[-1])
# the counter passes print their own line (they may run at a reduced sample count): rays of THAT frame
rays_of_pass = {}
for tag in ("fetch", "write"):
    line = json.loads(open(f"{root}/{tag}.json").read().strip().splitlines()[-1])
    rays_of_pass[tag] = line["value"] * 1e6 * line["ms_per_step"] / 1e3
work = bench["config"]["workload"]
m = re.search(r"(\d+)x(\d+) (\d+)spp (\d+) bounces", work)
m2 = re.search(r"(\d+) instanced triangles, (\d+) objects", work)
traffic = {"workload": work, "profile": os.path.basename(root.rstrip("/")), "resolution": [int(m.group(1)), int(m.group(2))],
           "bounces": int(m.group(4)), "scene": f"{m2.group(1)} triangles, {m2.group(2)} objects",
           "layout": "binary" if "binary nodes" in work else "wide",
           "rays_of_counter_pass": rays_of_pass, "kernels": {}}


def short(name):
    m = re.search(r"(\w+_kernel\d*(?:<[\w, ]+>)?)", name)
    return m.group(1) if m else name[:40]


print("== kernel trace statistics ==")
for f in glob.glob(root + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"{short(r['Name']):32s} calls {int(r['Calls']):7d} total_ms {float(r['TotalDurationNs'])/1e6:10.2f} "
              f"avg_us {float(r['AverageNs'])/1e3:9.1f} pct {float(r['Percentage']):6.2f}")

for tag, scale in (("fetch", 1.0), ("write", 1.0)):
    acc = collections.defaultdict(float)
    cnt = collections.defaultdict(int)
    for f in glob.glob(root + f"/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = (short(r["Kernel_Name"]), r["Counter_Name"])
            acc[k] += float(r["Counter_Value"])
            cnt[k] += 1
    print(f"== {tag} pass (KB -> bytes; streaming kernels: double the fetch figure) ==")
    for (k, c), v in sorted(acc.items()):
        print(f"{k:32s} {c:12s} dispatches {cnt[(k, c)]:7d} total_GB {v*1024*scale/1e9:10.3f} "
              f"per_launch_MB {v*1024*scale/cnt[(k, c)]/1e6:10.3f}")
        if "_kernel<false" in k or "_kernel7<false" in k or k.startswith("shade_") or k.startswith("generate_kernel"):
            # the counter passes render ONE frame: bytes / rays of that frame is the fabric traffic per ray
            traffic["kernels"].setdefault(k, {})[tag + "_bytes_per_ray"] = v * 1024 * scale / rays_of_pass[tag]
for f in glob.glob(root + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        if k in traffic["kernels"]:
            traffic["kernels"][k]["avg_launch_us"] = float(r["AverageNs"]) / 1e3
            traffic["kernels"][k]["share_of_gpu_time_pct"] = float(r["Percentage"])
traffic["config_key"] = f"{config}/{traffic['layout']}"
open(root + "/hbm_traffic.json", "w").write(json.dumps(traffic, indent=1) + "\n")
