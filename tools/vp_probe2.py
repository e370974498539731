#!/usr/bin/env python3
"""Dev probe: time the planned gather on the cfg-2 geometry (run with SGV3D_VP_CH / SGV3D_VP_MARGIN set to try window shapes)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import geometry_ref as G      # dev tool only
from sgv3d_amd.ops.voxel_pooling import VoxelPlan
geo = np.load(os.path.join(ROOT, "tests", "golden", "geometry.npz"))
n = "dair_p11_h5.5"
vs, vc, vn = G.voxel_params([0, 102.4, 0.4], [-51.2, 51.2, 0.4], [-5, 3, 8])
fr = G.create_frustum((864, 1536), 16, [-2.0, 0.0, 90])
gi, _ = G.geom_xyz_for_camera(fr, geo[f"{n}/sensor2ego"], geo[f"{n}/sensor2virtual"], geo[f"{n}/intrin"], geo[f"{n}/ida"],
                              geo[f"{n}/reference_height"], geo[f"{n}/bda"], vc, vs)
N = gi.shape[0] * gi.shape[1] * gi.shape[2]
X, Y, Z = (int(v) for v in vn)
C = int(os.environ.get("C", "80"))
g = torch.from_numpy(gi.reshape(1, N, 3)).cuda()
f = torch.randn(1, N, C, device="cuda")
out = torch.empty(1, Y, X, C, device="cuda")
plan = VoxelPlan(g, (X, Y, Z))
for _ in range(5):
    plan.pool(f, out)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
ev[0].record()
for i in range(40):
    plan.pool(f, out)
    ev[i + 1].record()
torch.cuda.synchronize()
ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(40))
alg = 12 * N + 4 * N * C + 4 * Y * X * C
print(f"CH={os.environ.get('SGV3D_VP_CH')} MARGIN={os.environ.get('SGV3D_VP_MARGIN')} median {ts[20]:.1f} us min {ts[0]:.1f} us  frac {alg / ts[20] / 8e6:.3f}", flush=True)
