#!/usr/bin/env python3
"""Per-launch table of one cfg-2 training step (forward + loss + backward; HIP events around every library launch, eager):
which layers the weight-gradient / data-gradient / BatchNorm time goes to.  BATCH=2 by default; DTYPE=bf16: the mixed-precision step."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgv3d_amd import hip_ops, synthetic
from sgv3d_amd.models.bev_height import BEVHeight

batch = int(os.environ.get("BATCH", "2"))
if os.environ.get("DTYPE", "f32") == "bf16":          # the mixed-precision step (tools/train_bench.py --dtype bf16)
    hip_ops.MFMA_BF16, hip_ops.BF16_ACTIVATIONS = True, False
dev = torch.device("cuda", 0)
bconf, hconf = synthetic.r50_256_conf()
torch.manual_seed(0)
model = BEVHeight(bconf, hconf).to(dev).train()
imgs = synthetic.make_images(batch, final=bconf['final_dim'], device=dev, seed=0)
mats = synthetic.make_mats(batch, device=dev)
boxes, labels = synthetic.make_gt(batch, seed=0, n_range=(10, 40), stress=False)
boxes, labels = [b.to(dev) for b in boxes], [l.to(dev) for l in labels]


def step():
    for p in model.parameters():
        p.grad = None
    preds = model(imgs, mats)
    targets = model.get_targets(boxes, labels)
    loss = model.loss(targets, preds)
    loss.backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
hip_ops.PROFILE_DETAIL = True
hip_ops.PROFILE = []
step()
torch.cuda.synchronize()
recs = hip_ops.PROFILE
hip_ops.PROFILE = None
rows = [(r[0], r[1], r[2].elapsed_time(r[3]) * 1e3) for r in recs]
PEAK = 2.5e15 if os.environ.get("DTYPE", "f32") == "bf16" else 157.3e12
# floor of a launch under both roofs: algorithmic flops at the MFMA peak of the mode's dtype, algorithmic bytes at 8 TB/s (BatchNorm:
# 12 / 20 B per element are not recorded here -> no floor)
floors = [(r[0], r[2].elapsed_time(r[3]) * 1e3, max((r[1] or 0) / PEAK, (r[4] or 0) / 8e12) * 1e6, (r[4] or 0)) for r in recs]
tot = sum(r[2] for r in rows)
agg = {}
for n, f, us in rows:
    k = n.split('|')[0]
    a = agg.setdefault(k, [0.0, 0.0, 0])
    a[0] += us; a[1] += f or 0; a[2] += 1
print(f"total {tot / 1e3:.2f} ms in {len(rows)} launches")
for k, (us, f, n) in sorted(agg.items(), key=lambda x: -x[1][0])[:14]:
    print(f"{us / 1e3:8.2f} ms  n={n:4d}  {f / us / 1e6 if f else 0:6.1f} TF  {k}")
print("--- launches by time" + (" (" + os.environ["FILTER"] + ")" if os.environ.get("FILTER") else ""))
for n, f, us in sorted((r for r in rows if r[0].startswith(os.environ.get("FILTER", ""))), key=lambda r: -r[2])[:int(os.environ.get("TOP", "40"))]:
    print(f"{us:8.1f} us {(f or 0) / us / 1e6:6.1f} TF  {n}")

print("--- launches by time above their two-roof floor (flops at the MFMA peak / bytes at 8 TB/s; launches that record bytes)")
for n, us, fl, nb in sorted((f for f in floors if f[3] > 0), key=lambda f: -(f[1] - f[2]))[:int(os.environ.get("TOP", "40"))]:
    print(f"{us:8.1f} us  floor {fl:7.1f} us  {nb / us / 1e6:5.2f} TB/s  {n}")
