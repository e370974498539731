import csv, sys, collections
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "patch" not in k and "igemm" not in k and "head" not in k: continue
    key = (k[:60], r["Grid_Size"])
    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
    line = f"{key[0]} grid {key[1]}: cycles {cyc:.0f}"
    if cyc:
        line += f" mfma_util {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (cyc * 1024) * 100:.1f}%"
    for n, v in m.items():
        if n not in ("GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES"):
            line += f" {n}={v:.3g}"
    print(line)
