#!/usr/bin/env python3
"""Per-kernel VALU 'tax' from a rocprofv3 PMC pass (SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE):
on gfx950 f32 MFMAs and VALU instructions share the SIMD's lanes even across waves (tools/ubench/mfma_valu_overlap.hip), so a
frame costs  sum(MFMA cycles) + 4 x sum(non-MFMA VALU instructions)  SIMD-cycles.    python tools/valu_tax.py <counter csv> [steps]"""
import collections, csv, sys
path, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
per = collections.defaultdict(lambda: collections.defaultdict(float))
seen = collections.defaultdict(set)
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen[k]:
        seen[k].add(r["Dispatch_Id"])
        per[k]["_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
rows = []
for k, c in per.items():
    valu = c.get("SQ_INSTS_VALU", 0.0) - c.get("SQ_INSTS_MFMA", 0.0)
    mfma_cycles = 32 if k.startswith(("head_wino4_kernel", "conv_f4res_kernel")) else 64      # v_mfma_f32_16x16x4_f32: 8 passes
    rows.append((k, len(seen[k]) / steps, c["_ns"] / 1e3 / steps, c.get("SQ_INSTS_MFMA", 0.0) * mfma_cycles / 1024 / steps, valu * 4 / 1024 / steps))
tot_m = sum(r[3] for r in rows); tot_v = sum(r[4] for r in rows); tot_t = sum(r[2] for r in rows)
print(f"{'kernel':60s} {'n/frame':>7s} {'us/frame':>9s} {'MFMA kcyc/SIMD':>15s} {'VALU kcyc/SIMD':>15s}   (MFMA at 64 cycles each, the 16x16x4 kernels 32; VALU at 4)")
for k, n, us, m, v in sorted(rows, key=lambda r: -(r[3] + r[4]))[:40]:
    print(f"{k:60s} {n:7.1f} {us:9.1f} {m / 1e3:15.1f} {v / 1e3:15.1f}")
print(f"{'TOTAL':60s} {'':7s} {tot_t:9.1f} {tot_m / 1e3:15.1f} {tot_v / 1e3:15.1f}   -> VALU / MFMA = {tot_v / max(tot_m, 1):.2f}; "
      f"at 2.0 GHz: MFMA {tot_m / 2e3:.0f} us + VALU {tot_v / 2e3:.0f} us per frame")
