#!/usr/bin/env python3
"""Run one conv shape a few times per tile (for rocprofv3 --pmc / --kernel-trace probes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgv3d_amd.hip_ops import PackedConv
import sgv3d_amd.hip_ops as H
H.AUTOTUNE = False
shape = os.environ.get("SHAPE", "1,512,54,96,512,3,1,1,1")
B, cin, Hh, W, cout, k, s, p, d = (int(v) for v in shape.split(","))
tiles = [int(t) for t in os.environ.get("TILES", "1,4").split(",")]
x = torch.randn(B, Hh, W, cin, device="cuda")
w = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
for t in tiles:
    conv = PackedConv(w, stride=s, pad=p, dil=d, relu=True, tile=t)
    out = None
    for _ in range(int(os.environ.get("REPS", "5"))):
        out = conv(x, out)
torch.cuda.synchronize()
print("done")
