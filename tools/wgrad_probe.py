#!/usr/bin/env python3
"""Run the weight-gradient kernel of one conv shape a few times per tile (for rocprofv3 --pmc probes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgv3d_amd import conv_grad
shape = os.environ.get("SHAPE", "1,512,54,96,512,3,1,1,1")
B, cin, H, W, cout, k, s, p, d = (int(v) for v in shape.split(","))
x = torch.randn(B, H, W, cin, device="cuda")
oh, ow = (H + 2 * p - d * (k - 1) - 1) // s + 1, (W + 2 * p - d * (k - 1) - 1) // s + 1
dy = torch.randn(B, oh, ow, cout, device="cuda")
for t in [int(v) for v in os.environ.get("TILES", "1,4").split(",")]:
    for _ in range(int(os.environ.get("REPS", "5"))):
        conv_grad.conv2d_backward_weight(x, dy, k, s, p, d, tile=t, split=int(os.environ.get("SPLIT", "0")))
torch.cuda.synchronize()
print("done")
