#!/usr/bin/env python3
"""Forward / data-gradient / weight-gradient time of cfg-2 layer shapes.  SHAPES="B,cin,H,W,cout,k,s,p,d;..." """
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgv3d_amd import conv_grad
from sgv3d_amd.hip_ops import PackedConv

DEFAULT = ("1,64,216,384,64,3,1,1,1;1,64,216,384,256,1,1,0,1;1,256,216,384,64,1,1,0,1;1,128,108,192,128,3,1,1,1;"
           "1,256,54,96,256,3,1,1,1;1,512,27,48,512,3,1,1,1;1,512,54,96,512,3,1,1,1;1,1024,54,96,256,1,1,0,1;"
           "1,128,216,384,128,3,2,1,1;1,160,128,128,160,3,1,1,1;1,640,32,32,640,3,1,1,1;1,256,256,256,64,3,1,1,1")


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n))[n // 3]


for shape in os.environ.get("SHAPES", DEFAULT).split(";"):
    B, cin, H, W, cout, k, s, p, d = (int(v) for v in shape.split(","))
    x = torch.randn(B, H, W, cin, device="cuda")
    w = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
    conv = PackedConv(w, stride=s, pad=p, dil=d)
    y = conv(x)
    dy = torch.randn_like(y)
    flops = 2.0 * y.numel() * cin * k * k
    t_f = timed(lambda: conv(x, y))
    conv_grad.conv2d_backward_data(dy, w, (H, W), s, p, d)          # packs + autotunes once
    t_d = timed(lambda: conv_grad.conv2d_backward_data(dy, w, (H, W), s, p, d), 5)
    res = [f"{shape:34s} fwd {t_f:7.1f} us {flops / t_f / 1e6:6.1f} TF | dgrad(+pack) {t_d:7.1f} us {flops / t_d / 1e6:6.1f} TF | wgrad"]
    for tile in (0,) + tuple(int(v) for v in os.environ.get("TILES", "").split(",") if v):
        for split in (0,) + tuple(int(v) for v in os.environ.get("SPLITS", "").split(",") if v):
            t_w = timed(lambda: conv_grad.conv2d_backward_weight(x, dy, k, s, p, d, split=split, tile=tile))
            res.append(f"t{tile} s{split}: {t_w:6.1f} us {flops / t_w / 1e6:5.1f} TF")
    print(" ".join(res), flush=True)
