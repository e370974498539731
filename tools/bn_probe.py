#!/usr/bin/env python3
"""Training-mode BatchNorm kernels (csrc/bn_train.hip) against their HBM floor on the shapes of a cfg-2 batch-2 step:
forward = statistics pass (read x) + apply pass (read x, write y) = 12 B per element; backward of relu(bn(x)) = reduce pass
(read x, dy) + apply pass (read x, dy, write dx) = 20 B per element; with a residual the backward also reads y twice and writes
the residual's gradient (32 B).  Prints microseconds and TB/s per call."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgv3d_amd.norm_grad import batch_norm_act                # noqa: E402

SHAPES = [(2, 432, 768, 64), (2, 216, 384, 64), (2, 216, 384, 256), (2, 108, 192, 128), (2, 108, 192, 512), (2, 54, 96, 256),
          (2, 54, 96, 1024), (2, 54, 96, 512), (2, 27, 48, 2048), (2, 256, 256, 64), (2, 128, 128, 160), (2, 64, 64, 320)]


from tools.vp_probe3 import graph_us                     # GPU time of a hipGraph of launches (not the host's launch cadence)

for shape in SHAPES:
    n = 1
    for v in shape:
        n *= v
    bn = torch.nn.BatchNorm2d(shape[-1]).cuda().train()
    x = torch.randn(*shape, device='cuda').requires_grad_(True)
    res = torch.randn(*shape, device='cuda').requires_grad_(True)
    dy = torch.randn(*shape, device='cuda')
    row = f"{'x'.join(map(str, shape)):>18}"
    for name, r, fb, bb in (("relu", None, 12, 20), ("res+relu", res, 16, 32)):
        def fwd():
            with torch.no_grad():
                batch_norm_act(bn, x, r, relu=True)
        t_f = graph_us(fwd, reps=5)
        def both():                # (forward + backward inside the capture: the autograd nodes must belong to the capturing stream)
            y = batch_norm_act(bn, x, r, relu=True)
            torch.autograd.grad(y, [x] + ([r] if r is not None else []), dy)
        t_b = graph_us(both, reps=5) - t_f
        row += f" | {name}: fwd {t_f:6.1f} us {n * fb / t_f / 1e6:5.2f} TB/s, bwd {t_b:6.1f} us {n * bb / t_b / 1e6:5.2f} TB/s"
    print(row, flush=True)
