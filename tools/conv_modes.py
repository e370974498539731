#!/usr/bin/env python3
"""One table per conv shape: best (tile, split) time of the f32-MFMA, bf16 and f32x3 variants of the implicit GEMM.
SHAPES="B,cin,H,W,cout,k,s,p,d;..."  (defaults: the cfg-2 layers the implicit GEMM serves)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgv3d_amd import hip_ops
from sgv3d_amd.hip_ops import PackedConv

DEFAULT = ("1,64,216,384,256,1,1,0,1;1,256,216,384,64,1,1,0,1;1,128,108,192,512,1,1,0,1;1,512,108,192,128,1,1,0,1;"
           "1,256,54,96,1024,1,1,0,1;1,1024,54,96,256,1,1,0,1;1,512,54,96,512,3,1,6,6;1,2560,54,96,512,1,1,0,1;"
           "1,128,216,384,128,3,2,1,1;1,512,27,48,2048,1,1,0,1")
shapes = os.environ.get("SHAPES", DEFAULT).split(";")
modes = os.environ.get("MODES", "f32,bf16,f32x3").split(",")
for shape in shapes:
    B, cin, Hh, W, cout, k, s, p, d = (int(v) for v in shape.split(","))
    x = torch.randn(B, Hh, W, cin, device="cuda")
    w = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
    res = torch.randn(B, (Hh + 2 * p - d * (k - 1) - 1) // s + 1, (W + 2 * p - d * (k - 1) - 1) // s + 1, cout, device="cuda")
    conv = PackedConv(w, stride=s, pad=p, dil=d, relu=True)
    flops = 2.0 * res.numel() * cin * k * k
    row = []
    for mode in modes:
        hip_ops.MFMA_BF16, hip_ops.MFMA_F32X3 = mode == "bf16", mode == "f32x3"
        best = None
        for t in (1, 2, 3, 4):
            for sk in (1, 2, 4):
                if sk > 1 and (conv.k_pad // 32 < 4 * sk or res.numel() // cout > 40000):
                    continue
                out = conv(x, tile=t, split_k=sk, residual=res)
                torch.cuda.synchronize()
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
                ev[0].record()
                for i in range(8):
                    conv(x, out, tile=t, split_k=sk, residual=res)
                    ev[i + 1].record()
                torch.cuda.synchronize()
                us = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(8))[2]
                if best is None or us < best[0]:
                    best = (us, t, sk)
        row.append(f"{mode} {best[0]:7.1f} us {flops / best[0] / 1e6:6.1f} TF (tile {best[1]} split {best[2]})")
    hip_ops.MFMA_BF16 = hip_ops.MFMA_F32X3 = False
    print(f"{shape:32s} | " + " | ".join(row), flush=True)
