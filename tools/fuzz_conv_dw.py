#!/usr/bin/env python3
"""Random-shape parity fuzz of the bf16 direct-weight kernel (sgv3d_conv_dw_bf16_forward, host tiles 31-35, NORMAL and DECONV) and
of the fused pair (sgv3d_conv_dw_bf16_pair_forward) against torch CPU fp64 on the bf16-rounded operands: kernel sizes 1-7,
stride, dilation, padding beyond the kernel, channel counts with odd numbers of 32-channel blocks, BN fold / residual / ReLU,
split-K,
channel-slice input and output with guard channels (written buffers are checked around the slice).
usage: fuzz_conv_dw.py [N=200] [seed=0]"""
import os, random, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgv3d_amd import hip_ops
from sgv3d_amd.hip_ops import PackedConv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hip_ops.MFMA_BF16 = True
rng = random.Random(seed)
g = torch.Generator().manual_seed(seed)
bad = launches = 0
for it in range(N):
    kind = rng.choice(["conv", "conv", "conv", "deconv", "pair"])
    B = rng.choice([1, 1, 2, 3])
    H, W = rng.randint(5, 36), rng.randint(5, 60)
    x_extra, y_extra = rng.choice([0, 0, 16]), rng.choice([0, 0, 16])
    x_coff, y_coff = x_extra // 2, y_extra // 2
    use_bn, use_res, use_relu = (rng.random() < 0.6 for _ in range(3))
    if kind == "deconv":
        ks = rng.choice([1, 2, 4])
        cin, cout = rng.choice([32, 64, 96, 160, 256]), rng.choice([8, 32, 64, 72])
        x = torch.randn(B, H, W, cin + x_extra, generator=g).bfloat16()
        w = torch.randn(cin, cout, ks, ks, generator=g) / cin ** 0.5
        sc = torch.rand(cout, generator=g) + 0.5 if use_bn else None
        sh = torch.randn(cout, generator=g) if use_bn else None
        ref = F.conv_transpose2d(x[..., x_coff:x_coff + cin].double().permute(0, 3, 1, 2), w.bfloat16().double(), stride=ks).permute(0, 2, 3, 1)
        if use_bn:
            ref = ref * sc.double() + sh.double()
        if use_relu:
            ref = ref.clamp_min(0)
        conv = PackedConv(w.cuda(), stride=ks, transposed=True, scale=None if sc is None else sc.cuda(), shift=None if sh is None else sh.cuda(),
                          relu=use_relu)
        res, desc = None, (kind, B, cin, H, W, cout, ks)
    else:
        k = rng.choice([1, 1, 3, 3, 3, 5, 7])
        stride = rng.choice([1, 1, 1, 2, 3])
        dil = rng.choice([1, 1, 2, 6]) if k > 1 else 1
        pad = rng.choice([0, (k // 2) * dil, (k // 2) * dil, (k // 2) * dil + 1])
        cin = rng.choice([32, 64, 96, 128, 160, 352])
        cout = 256 if kind == "pair" else rng.choice([8, 24, 64, 72, 136, 200, 264])
        if (H + 2 * pad - dil * (k - 1) - 1) < 0 or (W + 2 * pad - dil * (k - 1) - 1) < 0:
            continue
        x = torch.randn(B, H, W, cin + x_extra, generator=g).bfloat16()
        w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
        sc = torch.rand(cout, generator=g) + 0.5 if use_bn else None
        sh = torch.randn(cout, generator=g) if use_bn else None
        ref = F.conv2d(x[..., x_coff:x_coff + cin].double().permute(0, 3, 1, 2), w.bfloat16().double(), stride=stride, padding=pad, dilation=dil).permute(0, 2, 3, 1)
        if use_bn:
            ref = ref * sc.double() + sh.double()
        conv = PackedConv(w.cuda(), stride=stride, pad=pad, dil=dil, scale=None if sc is None else sc.cuda(), shift=None if sh is None else sh.cuda(),
                          relu=use_relu if kind == "conv" else True)
        desc = (kind, B, cin, H, W, cout, k, stride, pad, dil)
        if kind == "pair":
            cout2 = rng.choice([256, 512, 1024])
            w2 = torch.randn(cout2, 256, 1, 1, generator=g) / 16
            sc2, sh2 = torch.rand(cout2, generator=g) + 0.5, torch.randn(cout2, generator=g) * 0.3
            mid = ref.clamp_min(0).bfloat16().double()
            ref = F.conv2d(mid.permute(0, 3, 1, 2), w2.bfloat16().double()).permute(0, 2, 3, 1) * sc2.double() + sh2.double()
            conv2 = PackedConv(w2.cuda(), scale=sc2.cuda(), shift=sh2.cuda(), relu=use_relu)
            cout = cout2
        res = None
        if use_res:
            res = torch.randn(B, ref.shape[1], ref.shape[2], cout, generator=g).bfloat16()
            ref = ref + res.double()
        if use_relu:
            ref = ref.clamp_min(0)
    OH, OW = ref.shape[1], ref.shape[2]
    scale_ref = max(1.0, ref.abs().max().item())
    xd = x.cuda()
    for t in ((0,) if kind == "pair" else (31, 32, 33, 34, 35, 36, 37, 38, 39)):
        out = torch.full((B, OH, OW, cout + y_extra), -7.0, device="cuda", dtype=torch.bfloat16)
        try:
            if kind == "pair":
                if x_extra:
                    xd = x[..., x_coff:x_coff + cin].contiguous().cuda()
                o = hip_ops.conv_pair_bf16(conv, conv2, xd, None if res is None else res.cuda())
                out[..., y_coff:y_coff + cout] = o
            else:
                split = 1
                if kind == "conv" and t not in (36, 37, 39) and rng.random() < 0.5:                   # split-K: any count up to the layer's 64-k chunks
                    split = rng.randint(2, max(2, min(12, -(-(k * k * (cin // 32)) // 2))))
                    split = min(split, -(-(k * k * (cin // 32)) // 2))
                conv(xd, out, x_coff=x_coff, y_coff=y_coff, residual=None if res is None else res.cuda(), tile=t, split_k=split)
            launches += 1
        except Exception as e:
            bad += 1
            print("EXC", desc, t, str(e)[:160], flush=True)
            continue
        got = out.float().cpu().double()
        err = (got[..., y_coff:y_coff + cout] - ref).abs().max().item()
        guard_ok = (got[..., :y_coff] == -7).all() and (got[..., y_coff + cout:] == -7).all()
        tol = (2.0 ** -6 if kind == "pair" else 2.0 ** -8) * scale_ref
        if err > tol or not guard_ok or not torch.isfinite(got).all():
            bad += 1
            print("FAIL", desc, t, "split", split if kind == "conv" else 1, "err", err, "tol", tol, "guard", bool(guard_ok), dict(bn=use_bn, res=use_res, relu=use_relu, x_coff=x_coff, y_coff=y_coff))
torch.cuda.synchronize()
print(f"fuzz done (direct-weight kernel): {N} shapes, {launches} launches, failures: {bad}")
