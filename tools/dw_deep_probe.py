#!/usr/bin/env python3
"""The two-chunks-ahead tiles of the bf16 direct-weight kernel (36 / 37) against the plain ones (31 / 32) and split-K on the layers of
cfg-5 at batch 1 (one 54x96 / 27x48 / 108x192 map: 20 - 330 workgroups per launch), alone and with three launches in flight."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgv3d_amd import hip_ops
hip_ops.MFMA_BF16 = True
DEV = "cuda:0"
SHAPES = [  # cin, H, W, cout, k, stride, pad, dil, residual
    (1024, 54, 96, 256, 1, 1, 0, 1, False), (256, 54, 96, 1024, 1, 1, 0, 1, True), (256, 54, 96, 256, 3, 1, 1, 1, False),
    (512, 108, 192, 128, 1, 1, 0, 1, False), (128, 108, 192, 512, 1, 1, 0, 1, True), (2048, 27, 48, 512, 1, 1, 0, 1, False),
    (512, 27, 48, 2048, 1, 1, 0, 1, True), (512, 27, 48, 512, 3, 1, 1, 1, False), (512, 54, 96, 512, 3, 1, 1, 1, False),
    (512, 54, 96, 512, 3, 1, 6, 6, False), (2560, 54, 96, 512, 1, 1, 0, 1, False), (352, 64, 64, 352, 3, 1, 1, 1, False),
    (704, 32, 32, 704, 3, 1, 1, 1, False), (1024, 54, 96, 2048, 1, 2, 0, 1, False),
]
STREAMS = [torch.cuda.Stream() for _ in range(3)]


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def time3(fn, n=12):
    fn(); torch.cuda.synchronize()
    cur = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(cur)
    for s in STREAMS:
        s.wait_event(e0)
        with torch.cuda.stream(s):
            for _ in range(n):
                fn()
        cur.wait_stream(s)
    e1.record(cur); e1.synchronize()
    return e0.elapsed_time(e1) / (3 * n) * 1e3


print(f"{'layer':40} {'wgs':>4} | {'t31':>6} {'(3x)':>6} | {'t36':>6} {'(3x)':>6} | {'31 best split':>16} {'(3x)':>6}")
for cin, H, W, cout, k, stride, pad, dil, with_res in SHAPES:
    w = torch.randn(cout, cin, k, k, device=DEV) / (cin * k * k) ** 0.5
    conv = hip_ops.PackedConv(w, stride=stride, pad=pad, dil=dil, scale=torch.ones(cout, device=DEV), shift=torch.zeros(cout, device=DEV), relu=True)
    oh, ow = conv.out_hw(H, W)
    x = torch.randn(1, H, W, cin, device=DEV).bfloat16()
    out = torch.empty(1, oh, ow, cout, dtype=torch.bfloat16, device=DEV)
    res = torch.randn(1, oh, ow, cout, device=DEV).bfloat16() if with_res else None
    run = lambda t, sk: (lambda: conv(x, out, residual=res, tile=t, split_k=sk))
    nch = -(-(k * k * cin // 32) // 2)
    wgs = -(-oh * ow // 64) * -(-cout // 256)
    best = None
    for sk in (2, 3, 4, 6, 8):
        if nch // sk >= 4 and wgs * sk <= 1024:
            t = timeit(run(31, sk))
            if best is None or t < best[1]:
                best = (sk, t)
    bs = f"{'-':>16} {'-':>6}" if best is None else f"{'x%d %.1f' % best:>16} {time3(run(31, best[0])):6.1f}"
    name = f"{cin}->{cout} k{k} s{stride} d{dil} @{H}x{W}{' +res' if with_res else ''}"
    print(f"{name:40} {wgs:4d} | {timeit(run(31, 1)):6.1f} {time3(run(31, 1)):6.1f} | {timeit(run(36, 1)):6.1f} {time3(run(36, 1)):6.1f} | {bs}", flush=True)
