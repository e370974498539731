#!/usr/bin/env python3
"""Voxel pooling probe at cfg-2 size on the synthetic DAIR-like geometry (for rocprofv3 runs)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import geometry_ref as G
from sgv3d_amd.ops.voxel_pooling import VoxelPlan
geo = np.load(os.path.join(ROOT, "tests", "golden", "geometry.npz"))
n = "dair_p11_h5.5"
grid = os.environ.get("GRID", "256")
step = 0.4 if grid == "256" else 0.8
vs, vc, vn = G.voxel_params([0, 102.4, step], [-51.2, 51.2, step], [-5, 3, 8])
if os.environ.get("CFG5"):      # SGV3D BSM frustum: stride 8, 180 height bins in [-2, 3.5)
    fr = G.create_frustum((864, 1536), 8, [-2.0, 3.5, 180])
else:
    fr = G.create_frustum((864, 1536), 16, [-2.0, 0.0, 90])
gi, _ = G.geom_xyz_for_camera(fr, geo[f"{n}/sensor2ego"], geo[f"{n}/sensor2virtual"], geo[f"{n}/intrin"],
                              geo[f"{n}/ida"], geo[f"{n}/reference_height"], geo[f"{n}/bda"], vc, vs)
N = gi.shape[0] * gi.shape[1] * gi.shape[2]
g = torch.from_numpy(gi.reshape(1, N, 3)).cuda()
C = int(os.environ.get("CH", "80"))
f = torch.randn(1, N, C, device="cuda")
X, Y, Z = (int(v) for v in vn)
out = torch.empty(1, Y, X, C, device="cuda")
inr = ((gi[..., 0] >= 0) & (gi[..., 0] < X) & (gi[..., 1] >= 0) & (gi[..., 1] < Y) & (gi[..., 2] >= 0) & (gi[..., 2] < Z))
cnt = np.bincount((gi[..., 1].astype(np.int64) * X + gi[..., 0])[inr], minlength=X * Y)
print("N", N, "in-range", inr.mean(), "hit voxels", (cnt > 0).sum(), "mean", cnt[cnt > 0].mean(), "max", cnt.max(),
      "n>64", (cnt > 64).sum(), "n>256", (cnt > 256).sum(), "n>1024", (cnt > 1024).sum(), "n>8192", (cnt > 8192).sum())
reps = int(os.environ.get("REPS", "10"))
for _ in range(reps):
    plan = VoxelPlan(g, (X, Y, Z))
    plan.pool(f, out)
torch.cuda.synchronize()
print("done")
