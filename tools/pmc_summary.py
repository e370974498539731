#!/usr/bin/env python3
"""Median per-kernel counter values from rocprofv3 --pmc csv output.  usage: pmc_summary.py DIR [kernel-substring]"""
import collections, csv, glob, sys
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "conv_"
for f in sorted(glob.glob(d + "/*_counter_collection.csv")):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub not in k or len(k) > 300:
            continue
        vals[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f.replace("counter_collection", "kernel_trace"))):
        k = r["Kernel_Name"]
        if sub in k and len(k) <= 300:
            dur[k[:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print(f.split("/")[-1])
    for k, c in vals.items():
        ds = sorted(dur[k])
        print("  ", k, "median us", ds[len(ds) // 2] / 1e3, "n", len(ds))
        for n, v in sorted(c.items()):
            print("      %-32s %.4g" % (n, sorted(v)[len(v) // 2]))
