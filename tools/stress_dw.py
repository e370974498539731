#!/usr/bin/env python3
"""Race / determinism soak of the bf16 direct-weight kernels: every tile incl. the *_DEEP / 64x128 ones and two split-K cases (and the fused pair) on a few shapes, 25 launches each with
other work in flight on a second stream, outputs compared bitwise with the first launch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgv3d_amd import hip_ops

hip_ops.MFMA_BF16 = True
DEV = "cuda:0"
side = torch.cuda.Stream()
noise_a = torch.randn(4096, 4096, device=DEV)
bad = 0
SHAPES = [(4, 256, 68, 120, 256, 3, 1, 1, 1), (2, 96, 128, 128, 160, 7, 2, 3, 1), (4, 1024, 34, 60, 256, 1, 1, 0, 1), (1, 512, 54, 96, 512, 3, 1, 6, 6),
          (4, 64, 136, 240, 256, 1, 1, 0, 1), (2, 160, 64, 64, 320, 3, 2, 1, 1)]
for B, cin, H, W, cout, k, s, p, d in SHAPES:
    conv = hip_ops.PackedConv(torch.randn(cout, cin, k, k, device=DEV) / (cin * k * k) ** 0.5, stride=s, pad=p, dil=d,
                              scale=torch.rand(cout, device=DEV) + 0.5, shift=torch.randn(cout, device=DEV), relu=True)
    x = torch.randn(B, H, W, cin, device=DEV).bfloat16()
    oh, ow = conv.out_hw(H, W)
    res = torch.randn(B, oh, ow, cout, device=DEV).bfloat16()
    for t, sk in ((31, 1), (32, 1), (33, 1), (34, 1), (35, 1), (36, 1), (37, 1), (38, 1), (39, 1), (31, 3), (38, 2)):   # (tile, split-K)
        sk = min(sk, -(-(k * k * (cin // 32)) // 2))                # (no more splits than 64-k chunks)
        first = conv(x, residual=res, tile=t, split_k=sk, out_dtype=torch.bfloat16).clone()
        for it in range(25):
            with torch.cuda.stream(side):
                noise_a @ noise_a                                  # keeps the chip busy next to the kernel under test
            y = conv(x, residual=res, tile=t, split_k=sk, out_dtype=torch.bfloat16)
            if not torch.equal(y, first):
                bad += 1
                print("MISMATCH", (B, cin, H, W, cout, k, s, p, d), t, it, float((y.float() - first.float()).abs().max()))
                break
ca = hip_ops.PackedConv(torch.randn(256, 256, 3, 3, device=DEV) / 48, pad=1, scale=torch.ones(256, device=DEV), shift=torch.zeros(256, device=DEV), relu=True)
cb = hip_ops.PackedConv(torch.randn(1024, 256, 1, 1, device=DEV) / 16, scale=torch.ones(1024, device=DEV), shift=torch.zeros(1024, device=DEV), relu=True)
x = torch.randn(4, 68, 120, 256, device=DEV).bfloat16()
res = torch.randn(4, 68, 120, 1024, device=DEV).bfloat16()
first = hip_ops.conv_pair_bf16(ca, cb, x, res).clone()
for it in range(40):
    with torch.cuda.stream(side):
        noise_a @ noise_a
    if not torch.equal(hip_ops.conv_pair_bf16(ca, cb, x, res), first):
        bad += 1
        print("MISMATCH pair", it)
        break
torch.cuda.synchronize()
print("stress done, mismatches:", bad)
