#!/usr/bin/env python3
"""Weight gradient of the 3x3 / stride-1 layers of a cfg-2 training step (batch 2): the per-tap kernel (best of its tiles
and splits, as the first-call measurement picks) against the all-taps kernel (tile 5) over a few splits."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgv3d_amd import conv_grad, _lib
import ctypes

SHAPES = [  # B, cin, H, W, cout
    (2, 64, 216, 384, 64), (2, 128, 108, 192, 128), (2, 256, 54, 96, 256), (2, 512, 27, 48, 512), (2, 512, 54, 96, 512),
    (2, 160, 128, 128, 160), (2, 320, 64, 64, 320), (2, 640, 32, 32, 640), (2, 256, 256, 256, 64), (2, 64, 256, 256, 64),
]


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    evs[0].record()
    for i in range(n):
        fn(); evs[i + 1].record()
    torch.cuda.synchronize()
    return min(evs[i].elapsed_time(evs[i + 1]) for i in range(n)) * 1e3


for B, cin, H, W, cout in SHAPES:
    x = torch.randn(B, H, W, cin, device="cuda")
    dy = torch.randn(B, H, W, cout, device="cuda")
    flops = 2.0 * B * H * W * cout * cin * 9
    best_old = None
    units = B * H * -(-W // 32)
    for t in (1, 2, 3, 4):
        for sp in (0, 8, 32, 128):
            try:
                us = timeit(lambda: conv_grad.conv2d_backward_weight(x, dy, 3, 1, 1, 1, tile=t, split=sp))
            except _lib.SGV3DError:
                continue
            if best_old is None or us < best_old[0]:
                best_old = (us, t, sp)
    res = []
    t2 = -(-cout // 64) * -(-cin // 64)
    for sp in sorted({1, max(1, 256 // t2), max(1, 512 // t2), max(1, 1024 // t2), max(1, 2048 // t2)}):
        sp = min(sp, units)
        us = timeit(lambda: conv_grad.conv2d_backward_weight(x, dy, 3, 1, 1, 1, tile=5, split=sp))
        res.append((us, sp))
    b5 = min(res)
    print(f"{B}x{H}x{W} {cin}->{cout}: per-tap {best_old[0]:7.1f} us ({flops / best_old[0] / 1e6:5.1f} TF, tile {best_old[1]} split {best_old[2]})   "
          f"all-taps {b5[0]:7.1f} us ({flops / b5[0] / 1e6:5.1f} TF, split {b5[1]})   " + " ".join(f"{sp}:{us:.0f}" for us, sp in res), flush=True)
