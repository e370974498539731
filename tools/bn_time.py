#!/usr/bin/env python3
"""Fused training BatchNorm (+ReLU, +residual) forward / backward time and effective HBM rate for cfg-2 layer shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgv3d_amd.norm_grad import batch_norm_act

def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n))[n // 3]

for shape, res in (((2, 216, 384, 256), True), ((2, 216, 384, 64), False), ((2, 108, 192, 512), True), ((2, 54, 96, 1024), True),
                   ((2, 27, 48, 2048), True), ((2, 128, 128, 160), False), ((2, 54, 96, 512), False)):
    x = torch.randn(shape, device="cuda").requires_grad_(True)
    r = torch.randn(shape, device="cuda").requires_grad_(True) if res else None
    bn = torch.nn.BatchNorm2d(shape[-1]).cuda().train()
    dy = torch.randn(shape, device="cuda")
    mb = x.numel() * 4 / 1e6
    t_f = timed(lambda: batch_norm_act(bn, x, r, True))
    y = batch_norm_act(bn, x, r, True)
    t_b = timed(lambda: torch.autograd.grad(y, [x] + ([r] if res else []) + [bn.weight, bn.bias], dy, retain_graph=True))
    pf, pb = (3 + (1 if res else 0)), (3 + 4 + (1 if res else 0))
    print(f"{str(shape):24s} res={res!s:5s} fwd {t_f:7.1f} us {pf * mb / t_f:6.2f} TB/s | bwd {t_b:7.1f} us {pb * mb / t_b:6.2f} TB/s")
