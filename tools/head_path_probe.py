#!/usr/bin/env python3
"""Dev probe: the 36 CenterHead branches at 256x256 -- fused kernel vs two-kernel path with each first-layer algorithm -- alone
and with three copies in flight."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgv3d_amd import hip_ops
from sgv3d_amd.hip_ops import PackedConv

counts = []
for nc in (1, 2, 2, 1, 2, 2):
    counts += [2, 1, 3, 2, 2, nc]
nb, total = len(counts), sum(counts)
B, H, W, cin = 1, 256, 256, 64
x = torch.randn(B, H, W, cin, device="cuda")
w1 = torch.randn(nb * 64, cin, 3, 3, device="cuda") / 24.0
sc, sh = torch.rand(nb * 64, device="cuda") + 0.5, torch.randn(nb * 64, device="cuda") * 0.2
w2 = (torch.randn(total, 64, 3, 3, device="cuda") / 24.0).permute(0, 2, 3, 1).contiguous()
b2 = torch.randn(total, device="cuda")
first = PackedConv(w1, pad=1, scale=sc, shift=sh, relu=True)
ob = torch.tensor([0] + list(np.cumsum(counts)), dtype=torch.int32, device="cuda")
bo = torch.repeat_interleave(torch.arange(nb, dtype=torch.int32), torch.tensor(counts)).cuda()
hip_ops.TUNE_STREAMS = 3
paths = {"fused": lambda: hip_ops.centerhead_branches(x, first, w2, b2, ob, nb)}
for t in (9, 10, 15, 6, 5, 4):
    def mk(t=t):
        return lambda: hip_ops.head_final_conv(first(x, group_planes=64, tile=t, split_k=1), w2, b2, bo, nb, 64)
    paths[f"first tile {t} + final"] = mk()
for name, fn in paths.items():
    try:
        fn(); torch.cuda.synchronize()
        hip_ops.TUNE_STREAMS = 1
        alone = hip_ops.time_callable(fn, x.device, rounds=3) / 3
        hip_ops.TUNE_STREAMS = 3
        loaded = hip_ops.time_callable(fn, x.device, rounds=3) / 9
        print(f"{name:26s} {alone * 1e3:8.1f} us alone   {loaded * 1e3:8.1f} us per call with three in flight", flush=True)
    except Exception as e:
        print(name, "failed:", str(e)[:120], flush=True)
