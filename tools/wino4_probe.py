#!/usr/bin/env python3
"""Dev probe: F(4x4) (tiles 9 / 10) against the F(2x2) kernels (5 full, 8 half, with their split-K) and the implicit GEMM on
the many-channel 3x3 layers of cfg-2; us per launch alone and with three launches in flight."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgv3d_amd.hip_ops import PackedConv

LAYERS = [(512, 54, 96, 512), (256, 54, 96, 256), (512, 27, 48, 512), (640, 32, 32, 640), (320, 64, 64, 320), (128, 108, 192, 128),
          (160, 128, 128, 160)]
CANDS = [(9, 1), (10, 1), (15, 1), (8, 1), (8, 2), (5, 1), (5, 2), (5, 3), (4, 1)]
streams = [torch.cuda.Stream() for _ in range(3)]


def time_alone(fn, n=12):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n))[n // 2]


def time_loaded(fn, n=8):
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


DIL = [(512, 54, 96, 512, 6), (512, 54, 96, 512, 12), (512, 54, 96, 512, 18)]
for layer in [l + (1,) for l in LAYERS] + DIL:
    cin, H, W, cout, dil = layer
    x = torch.randn(1, H, W, cin, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
    sc, sh = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda")
    conv = PackedConv(w, pad=dil, dil=dil, scale=sc, shift=sh, relu=True)
    out = torch.empty(1, H, W, cout, device="cuda")
    line = f"{cin:4d}->{cout:4d} @{H}x{W} d{dil}: "
    for t, sk in CANDS:
        if t in (5, 8) and conv.w_wino is None:
            continue
        if t in (9, 10, 15) and not conv.wino4_ok():
            continue
        if t in (5, 8) and cin // 4 // sk < 8 and sk > 1:
            continue
        fn = lambda: conv(x, out, tile=t, split_k=sk)
        try:
            line += f" [{t}.{sk}] {time_alone(fn):6.1f}/{time_loaded(fn):6.1f}"
        except Exception as e:
            line += f" [{t}.{sk}] fail"
    print(line, flush=True)
