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
fr = G.create_frustum((864, 1536), 16, [-2.0, 0.0, 90])
gi, _ = G.geom_xyz_for_camera(fr, geo[f"{n}/sensor2ego"], geo[f"{n}/sensor2virtual"], geo[f"{n}/intrin"],
                              geo[f"{n}/ida"], geo[f"{n}/reference_height"], geo[f"{n}/bda"], vc, vs)
N = gi.shape[0] * gi.shape[1] * gi.shape[2]
g = torch.from_numpy(gi.reshape(1, N, 3)).cuda()
f = torch.randn(1, N, 80, device="cuda")
X, Y, Z = (int(v) for v in vn)
out = torch.empty(1, Y, X, 80, device="cuda")
reps = int(os.environ.get("REPS", "10"))
for _ in range(reps):
    plan = VoxelPlan(g, (X, Y, Z))
    plan.pool(f, out)
torch.cuda.synchronize()
print("done")
