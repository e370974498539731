#!/usr/bin/env python3
"""One training step out of a rocprofv3 kernel trace (tools/train_trace.sh): wall time, GPU-busy time, launches, the idle gaps
(which kernel the GPU waited for) and the kernels by total time.  usage: trace_step_gaps.py <kernel_trace.csv>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
opt = [i for i, k in enumerate(ks) if 'adamw' in k[2].lower()]
steps, prev = [], None
for i in opt:
    if prev is None or ks[i][0] - prev > 20e6:
        steps.append(i)
    prev = ks[i][0]
a, b = steps[-2], steps[-1]
win = ks[a:b]
t0, t1 = win[0][0], win[-1][1]
busy, cur_s, cur_e = 0, None, None
gaps = []
for s, e, n in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"step window {(t1 - t0) / 1e6:.2f} ms, GPU busy {busy / 1e6:.2f} ms, {len(win)} launches, {len(gaps)} gaps = {sum(g for g, _ in gaps) / 1e6:.2f} ms")
hist = collections.Counter()
for g, _ in gaps:
    hist[min(int(g / 1e3) // 5 * 5, 100)] += g
print("idle time by gap length (us bucket: ms):", {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})
byk = collections.defaultdict(lambda: [0, 0])
for g, n in gaps:
    if g > 10e3:
        byk[n[:90]][0] += g
        byk[n[:90]][1] += 1
print("gaps > 10 us, by the kernel that followed:")
for n, (g, c) in sorted(byk.items(), key=lambda x: -x[1][0])[:15]:
    print(f"  {g / 1e6:7.2f} ms n={c:4d} {n}")
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in win:
    agg[n][0] += e - s
    agg[n][1] += 1
print("kernels by time:")
for n, (t, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"  {t / 1e6:7.2f} ms n={c:4d} {n[:120]}")
