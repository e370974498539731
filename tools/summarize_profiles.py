#!/usr/bin/env python3
"""Turn the raw rocprofv3 CSVs of tools/profile_round.sh into the small files kept under profiles/.

    python tools/summarize_profiles.py gpurun_out/profiles_r01 r01
"""
import collections
import csv
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0]


def pmc_per_kernel(path, counter):
    agg = collections.defaultdict(list)
    if not os.path.exists(path):
        return {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


for f in (f"{tag}_bench.json", f"{tag}_bench_kernel_stats.csv", f"{tag}_vp_kernel_stats.csv",
          f"{tag}_voxel_pooling_microbench.json", f"{tag}_bench_under_rocprof.json"):
    p = os.path.join(src, f)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f))

traffic = {}
for prefix, key in ((f"{tag}_pmc", "bench"), (f"{tag}_vp_pmc", "vp_probe")):
    fetch = pmc_per_kernel(os.path.join(src, f"{prefix}_fetch_counter_collection.csv"), "FETCH_SIZE")
    write = pmc_per_kernel(os.path.join(src, f"{prefix}_write_counter_collection.csv"), "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f_kb, nf = fetch.get(k, (0.0, 0))
        w_kb, nw = write.get(k, (0.0, 0))
        # MI355X_MICROARCH.md §HBM: counters are in KB; on gfx950 FETCH_SIZE reports exactly half the
        # bytes of wide coalesced reads -> doubled; WRITE_SIZE is exact for 16-B stores / float atomics.
        out[k] = {"fetch_bytes_per_launch": 2.0 * f_kb * 1024, "write_bytes_per_launch": w_kb * 1024,
                  "hbm_bytes_per_launch": (2.0 * f_kb + w_kb) * 1024, "launches": max(nf, nw),
                  "raw_FETCH_SIZE_KB": f_kb, "raw_WRITE_SIZE_KB": w_kb}
    traffic[key] = out
json.dump(traffic, open(os.path.join(dst, f"{tag}_hbm_traffic.json"), "w"), indent=1)
print("wrote", os.path.join(dst, f"{tag}_hbm_traffic.json"))
for key in traffic:
    for k, v in sorted(traffic[key].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]:
        print(f"{key:9s} {k[:60]:60s} {v['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch  (n={v['launches']})")
