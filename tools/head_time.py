#!/usr/bin/env python3
"""Time the CenterHead branch layers at cfg-2 size: fused kernel vs (Winograd first layer + head_final_conv)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgv3d_amd import hip_ops
from sgv3d_amd.hip_ops import PackedConv

counts = [2, 1, 3, 2, 2, 1] * 6
nb, total = len(counts), sum(counts)
H = W = int(os.environ.get("HW", "256"))
x = torch.randn(1, H, W, 64, device="cuda")
w1 = torch.randn(nb * 64, 64, 3, 3, device="cuda") / 24
first = PackedConv(w1, pad=1, scale=torch.rand(nb * 64, device="cuda") + 0.5, shift=torch.randn(nb * 64, device="cuda") * 0.1, relu=True)
w2 = (torch.randn(total, 3, 3, 64, device="cuda") / 24).contiguous()
b2 = torch.randn(total, device="cuda")
ob = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32, device="cuda")
boo = torch.tensor(sum(([i] * c for i, c in enumerate(counts)), []), dtype=torch.int32, device="cuda")


def timeit(fn, n=6):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n))[n // 3]


out = torch.empty(1, total, H, W, device="cuda")
print("fused                 %9.1f us" % timeit(lambda: hip_ops.centerhead_branches(x, first, w2, b2, ob, nb, out)))
hid = first(x, group_planes=64, tile=6, split_k=1)
print("winograd first layer  %9.1f us" % timeit(lambda: first(x, hid, group_planes=64, tile=6, split_k=1)))
print("head_final_conv       %9.1f us" % timeit(lambda: hip_ops.head_final_conv(hid, w2, b2, boo, nb, 64, out)))
