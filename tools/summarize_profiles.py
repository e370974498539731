#!/usr/bin/env python3
"""Turn the raw rocprofv3 CSVs of tools/profile_round.sh into the small files kept under profiles/.

    python tools/summarize_profiles.py <dir with the raw CSVs> r01 [<destination>, default profiles/]
"""
import collections
import csv
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles")
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
          f"{tag}_voxel_pooling_microbench.json", f"{tag}_bench_under_rocprof.json",
          f"{tag}_bench_3inflight_kernel_stats.csv", f"{tag}_bench_3inflight_under_rocprof.json",
          f"{tag}_layers_cfg2_tiles_for_3inflight.txt", f"{tag}_layers_cfg2_tiles_for_1inflight.txt",
          f"{tag}_wino4_kernel_stats.csv", f"{tag}_bench_cfg3_bf16_b4.json", f"{tag}_bench_cfg5_bf16.json",
          f"{tag}_bench_cfg5_bf16_b4.json", f"{tag}_bench_cfg3_fp32_b4.json", f"{tag}_bench_cfg5.json",
          f"{tag}_bench_cfg3_bf16_kernel_stats.csv", f"{tag}_harness_phases.txt", f"{tag}_harness_kernel_stats.csv",
          f"{tag}_gather_probe.txt", f"{tag}_bench_roofline_kernel_stats.csv", f"{tag}_bench_roofline_under_rocprof.json",
          f"{tag}_bench_cfg5_bf16_kernel_stats.csv", f"{tag}_bench_cfg3_bf16_under_rocprof.json", f"{tag}_bench_cfg5_bf16_under_rocprof.json",
          f"{tag}_layers_cfg3_bf16_b4.txt", f"{tag}_layers_cfg5_bf16_b1.txt"):
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

# MFMA utilisation per kernel (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES ... of `bench.py --streams 1`):
#   MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs x 1024 SIMDs)   (busy cycles of the matrix
#   pipes over the kernel's own cycles; rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs)
#   executed TFLOP/s = SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 / kernel duration; clock = GRBM_GUI_ACTIVE/8/duration
mfma_csv = os.path.join(src, f"{tag}_pmc_mfma_counter_collection.csv")
if os.path.exists(mfma_csv):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(mfma_csv)):
        k = short(r["Kernel_Name"])
        if not k.startswith(("conv_", "head_wino4_kernel", "head_bf16", "gemm16_grouped", "gemm_x3_grouped", "dcn3x3")):      # the MFMA kernels
            continue
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen[k]:
            seen[k].add(r["Dispatch_Id"])
            per[k]["_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    out = {}
    for k, c in per.items():
        if not c.get("GRBM_GUI_ACTIVE") or not c["_ns"]:
            continue
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        out[k] = {"launches": len(seen[k]), "time_ms": c["_ns"] / 1e6,
                  "mfma_util": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024),
                  "executed_tflops": c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512 / (c["_ns"] * 1e-9) / 1e12,
                  "clock_ghz": cyc / c["_ns"],
                  "executed_frac_of_157.3": c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512 / (c["_ns"] * 1e-9) / 1e12 / 157.3}
    json.dump(out, open(os.path.join(dst, f"{tag}_mfma_util.json"), "w"), indent=1)
    print("wrote", os.path.join(dst, f"{tag}_mfma_util.json"))
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["time_ms"]):
        print(f"mfma      {k[:50]:50s} util {v['mfma_util'] * 100:5.1f} %  {v['executed_tflops']:6.1f} TF executed  "
              f"clock {v['clock_ghz']:.2f} GHz  ({v['time_ms']:.2f} ms over {v['launches']} launches)")
print("wrote", os.path.join(dst, f"{tag}_hbm_traffic.json"))
for key in traffic:
    for k, v in sorted(traffic[key].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]:
        print(f"{key:9s} {k[:60]:60s} {v['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch  (n={v['launches']})")
