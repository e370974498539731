#!/usr/bin/env python3
"""Time one conv shape over (tile, split) candidates.  SHAPE=B,cin,H,W,cout,k,s,p,d  CANDS=tile:split,..."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgv3d_amd.hip_ops import PackedConv

shapes = os.environ.get("SHAPES", "1,512,54,96,512,3,1,1,1;1,64,256,256,2304,3,1,1,1").split(";")
cands = [tuple(int(v) for v in c.split(":")) for c in os.environ.get("CANDS", "4:1,3:3,5:1,5:2,5:3,5:4").split(",")]
for shape in shapes:
    B, cin, Hh, W, cout, k, s, p, d = (int(v) for v in shape.split(","))
    x = torch.randn(B, Hh, W, cin, device="cuda")
    w = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
    conv = PackedConv(w, stride=s, pad=p, dil=d, relu=True)
    flops = 2.0 * B * Hh * W // (s * s) * cout * cin * k * k
    for t, sk in cands:
        if t == 5 and (conv.w_wino is None or cin // 8 < sk):
            continue
        out = conv(x, tile=t, split_k=sk)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
        ev[0].record()
        for i in range(10):
            conv(x, out, tile=t, split_k=sk)
            ev[i + 1].record()
        torch.cuda.synchronize()
        us = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(10))[3]
        print(f"{shape:34s} tile {t} split {sk}: {us:8.1f} us  {flops / us / 1e6:7.1f} TF(direct-equivalent)", flush=True)
