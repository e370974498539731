"""Winograd: the full kernel (one workgroup per CU) against the half-position variant (two per CU), per layer shape of cfg-2,
best split-K of each, alone and with three copies in flight on three streams."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgv3d_amd import hip_ops

DEV = "cuda:0"
SHAPES = [(1, 512, 54, 96, 512), (1, 160, 128, 128, 160), (1, 320, 64, 64, 320), (1, 640, 32, 32, 640), (1, 256, 256, 256, 64),
          (1, 64, 216, 384, 64), (1, 128, 108, 192, 128), (1, 256, 54, 96, 256), (1, 512, 27, 48, 512)]
streams = [torch.cuda.Stream() for _ in range(3)]


def timeit(fn, n=8, multi=False):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream()
    e0.record()
    if multi:
        for s in streams:
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                for _ in range(n):
                    fn()
        for s in streams:
            cur.wait_stream(s)
    else:
        for _ in range(n):
            fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3 / (3 if multi else 1)


for B, cin, H, W, cout in SHAPES:
    w = torch.randn(cout, cin, 3, 3, device=DEV) / (cin * 9) ** 0.5
    conv = hip_ops.PackedConv(w, pad=1, scale=torch.ones(cout, device=DEV), shift=torch.zeros(cout, device=DEV), relu=True)
    x = torch.randn(B, H, W, cin, device=DEV)
    outs = [torch.empty(B, H, W, cout, device=DEV) for _ in range(1)]
    line = f"{H}x{W} {cin}->{cout}:"
    for name, t in (("full", hip_ops.TILE_WINO), ("half", hip_ops.TILE_WINO_HALF)):
        best = (1e9, 0); bestm = (1e9, 0)
        for sk in (1, 2, 3, 4, 6):
            if cin // 8 // sk < 2:
                continue
            us = timeit(lambda: conv(x, outs[0], tile=t, split_k=sk))
            best = min(best, (us, sk))
            usm = timeit(lambda: conv(x, outs[0], tile=t, split_k=sk), multi=True)
            bestm = min(bestm, (usm, sk))
        line += f"  {name}: alone {best[0]:6.1f} us (split {best[1]}), 3 in flight {bestm[0]:6.1f} us per launch (split {bestm[1]});"
    print(line, flush=True)
