#!/usr/bin/env python3
"""Per-launch time / TFLOP/s table of one cfg-2 forward (HIP events around every launch)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgv3d_amd import hip_ops, synthetic as S
from sgv3d_amd.models.bev_height import BEVHeight

cfg = os.environ.get("CONFIG", "cfg2")
batch = int(os.environ.get("BATCH", "1"))
hip_ops.MFMA_BF16 = os.environ.get("DTYPE", "f32") == "bf16"
bc, hc = {"cfg2": S.r50_256_conf, "cfg3": S.r101_512_conf, "cfg5": S.bsm_r101_256_conf}[cfg]()
torch.manual_seed(0)
m = BEVHeight(bc, hc).eval()
S.randomize_norm_stats_(m, 0, residual_gamma=0.3)
m = m.cuda()
imgs = S.make_images(batch, bc['final_dim'], device='cuda')
mats = S.make_mats(batch, device='cuda')
with torch.no_grad():
    for _ in range(3):
        m(imgs, mats)
    torch.cuda.synchronize()
    hip_ops.PROFILE_DETAIL = True
    reps = 5
    hip_ops.PROFILE = []
    for _ in range(reps):
        m(imgs, mats)
    torch.cuda.synchronize()
recs = hip_ops.PROFILE
hip_ops.PROFILE = None
n = len(recs) // reps
rows = []
for i in range(n):
    name, flops = recs[i][0], recs[i][1]
    us = sorted(recs[i + r * n][2].elapsed_time(recs[i + r * n][3]) * 1e3 for r in range(reps))[reps // 2]
    rows.append((i, name, flops, us))
tot = sum(r[3] for r in rows)
print(f"{'#':>3} {'us':>8} {'TF':>6} {'%pk':>5}  kernel")
for i, name, flops, us in rows:
    tf = flops / us / 1e6 if flops else 0
    print(f"{i:3d} {us:8.1f} {tf:6.1f} {tf / (2500.0 if hip_ops.MFMA_BF16 else 157.3) * 100:5.1f}  {name}")
print("total us", tot)
# launches ranked by time above their two-roof floor: algorithmic flops at the mode's MFMA peak, algorithmic bytes at 8 TB/s (the byte
# count of a record assumes f32 tensors; with bf16 activations in HBM the true floor is up to 2x lower)
PEAK = 2.5e15 if hip_ops.MFMA_BF16 else 157.3e12
ex = []
for i, name, flops, us in rows:
    nb = recs[i][4] or 0
    fl = max((flops or 0) / PEAK, nb / 8e12) * 1e6
    ex.append((us - fl, us, fl, nb, name))
print("--- by time above the two-roof floor")
for d, us, fl, nb, name in sorted(ex, reverse=True)[:int(os.environ.get("TOP", "30"))]:
    print(f"{us:8.1f} us  floor {fl:7.1f} us  {nb / us / 1e6 if us else 0:5.2f} TB/s  {name}")
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "layers.json"), "w"))
hip_ops.save_tune_db()          # no-op unless SGV3D_TUNE_CACHE is set (tools/layer_traffic.py replays the same choices under rocprofv3)
