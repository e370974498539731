#!/usr/bin/env python3
"""Dev probe: the ResNet-50 1x1 layers of cfg-2 on the 64x64 tile with the swapped-operand (16-byte) epilogue and without
(SGV3D_NO_PW_KERNEL / SGV3D_SWAP_EPI toggled in-process), residual + ReLU as in the bottlenecks.  Prints us per launch, alone and with three
concurrent copies (what the frame pipeline runs)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgv3d_amd.hip_ops import PackedConv

LAYERS = [(64, 216, 384, 256, True), (256, 216, 384, 64, False), (128, 108, 192, 512, True), (512, 108, 192, 128, False),
          (256, 54, 96, 1024, True), (1024, 54, 96, 256, False), (512, 27, 48, 2048, True), (64, 216, 384, 64, False)]
streams = [torch.cuda.Stream() for _ in range(3)]


def time_alone(fn, n=20):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n))[n // 2]


def time_loaded(fn, n=10):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cur = torch.cuda.current_stream()
    e0.record(cur)
    for s in streams:
        s.wait_event(e0)
        with torch.cuda.stream(s):
            for _ in range(n):
                fn()
        cur.wait_stream(s)
    e1.record(cur)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * len(streams))


for cin, H, W, cout, with_res in LAYERS:
    x = torch.randn(1, H, W, cin, device="cuda")
    w = torch.randn(cout, cin, 1, 1, device="cuda") / cin ** 0.5
    sc, sh = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda")
    conv = PackedConv(w, scale=sc, shift=sh, relu=True, tile=4)
    res = torch.randn(1, H, W, cout, device="cuda") if with_res else None
    out = torch.empty(1, H, W, cout, device="cuda")
    fn = lambda: conv(x, out, residual=res, tile=4, split_k=1)
    row = []
    time_loaded(fn)
    for var, flag in (("SGV3D_NO_PW_KERNEL", "1"), (None, None), ("SGV3D_SWAP_EPI", "1")):
        if var:
            os.environ[var] = flag
        try:
            row.append((time_alone(fn), time_loaded(fn)))
        finally:
            if var:
                os.environ.pop(var, None)
    print(f"{cin:5d}->{cout:5d} @{H}x{W} res={int(with_res)}:  generic {row[0][0]:6.1f} us alone / {row[0][1]:6.1f} loaded   "
          f"pointwise (default) {row[1][0]:6.1f} / {row[1][1]:6.1f}   swapped 16-byte epilogue {row[2][0]:6.1f} / {row[2][1]:6.1f}", flush=True)
