"""Per kernel symbol and grid size: mean of every counter of a rocprofv3 --pmc counter_collection.csv (+ MFMA utilisation)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2:] or ["patch", "igemm", "head", "wino"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if not any(w in k for w in want):
        continue
    acc[(k[:90], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
    line = f"{key[0]} grid {key[1]}:"
    if cyc:
        line += f" cycles {cyc:.0f} mfma_util {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (cyc * 1024) * 100:.1f}%"
    for n, v in sorted(m.items()):
        if n not in ("GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES"):
            line += f" {n}={v:.4g}"
    print(line)
