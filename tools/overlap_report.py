#!/usr/bin/env python3
"""How the kernels of the frames in flight overlap: from a rocprofv3 --kernel-trace csv of bench.py (three frames in flight), for the
last N ms of the trace: per kernel name the time it runs ALONE on the device, the time it shares with kernels of other streams, and the
share of the wall time in which exactly k kernels are running.   usage: overlap_report.py <kernel_trace.csv> [window_ms]"""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id") or r.get("Queue_Id")))
rows.sort()
window = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 80e6
t_end = max(r[1] for r in rows)
rows = [r for r in rows if r[0] >= t_end - window]
t0 = rows[0][0]
events = []
for i, (s, e, n, q) in enumerate(rows):
    events.append((s, 1, i))
    events.append((e, -1, i))
events.sort()
active = set()
alone = collections.Counter()
shared = collections.Counter()
level = collections.Counter()
prev = events[0][0]
for t, d, i in events:
    dt = t - prev
    if dt > 0:
        level[len(active)] += dt
        for j in active:
            (alone if len(active) == 1 else shared)[rows[j][2]] += dt
    prev = t
    if d > 0:
        active.add(i)
    else:
        active.discard(i)
wall = events[-1][0] - events[0][0]
print(f"window {wall / 1e6:.2f} ms, {len(rows)} kernels")
print("kernels running at once -> share of the wall time:", {k: round(v / wall, 3) for k, v in sorted(level.items())})
tot = collections.Counter()
for k in set(alone) | set(shared):
    tot[k] = alone[k] + shared[k]
print(f"{'kernel':70s} {'total ms':>9s} {'alone %':>8s} {'avg us':>8s}")
cnt = collections.Counter(r[2] for r in rows)
for k, v in tot.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 16):
    print(f"{k[:70]:70s} {v / 1e6:9.2f} {100 * alone[k] / v:8.1f} {v / cnt[k] / 1e3:8.1f}")
