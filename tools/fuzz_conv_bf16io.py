#!/usr/bin/env python3
"""Random-shape parity fuzz of the bf16-activation convolution entry points (sgv3d_conv2d_forward_bf16io, all tiles and split-K;
sgv3d_conv3x3_patch_bf16_forward with its cin split) against torch CPU fp64 on the bf16-rounded operands: every combination of
f32 / bf16 input and output, BN fold / residual / ReLU, channel-slice input and output with guard channels.
usage: fuzz_conv_bf16io.py [N=200] [seed=0]"""
import os, random, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgv3d_amd import hip_ops
from sgv3d_amd.hip_ops import PackedConv, TILE_PATCH

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hip_ops.MFMA_BF16 = True
rng = random.Random(seed)
g = torch.Generator().manual_seed(seed)
bad = launches = 0
for it in range(N):
    k = rng.choice([1, 3, 3, 3])
    stride = rng.choice([1, 1, 1, 2])
    dil = rng.choice([1, 1, 1, 2]) if k == 3 else 1
    pad = rng.choice([0, (k // 2) * dil, (k // 2) * dil, (k // 2) * dil])
    cin = rng.choice([8, 24, 32, 64, 96, 128, 160, 352])
    cout = rng.choice([8, 24, 40, 64, 72, 128, 136, 200])
    B = rng.choice([1, 1, 2, 3])
    H, W = rng.randint(7, 40), rng.randint(7, 70)
    if (H + 2 * pad - dil * (k - 1) - 1) < 0 or (W + 2 * pad - dil * (k - 1) - 1) < 0:
        continue
    x_extra, y_extra = rng.choice([0, 0, 16]), rng.choice([0, 0, 16])
    x = torch.randn(B, H, W, cin + x_extra, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    use_bn, use_res, use_relu = (rng.random() < 0.6 for _ in range(3))
    sc = torch.rand(cout, generator=g) + 0.5 if use_bn else None
    sh = torch.randn(cout, generator=g) if use_bn else None
    x_coff, y_coff = x_extra // 2, y_extra // 2
    xb, yb = rng.random() < 0.7, rng.random() < 0.7
    if not (xb or yb):
        xb = True
    xin = x[..., x_coff:x_coff + cin]
    ref = F.conv2d(xin.bfloat16().double().permute(0, 3, 1, 2), w.bfloat16().double(), stride=stride, padding=pad, dilation=dil).permute(0, 2, 3, 1)
    OH, OW = ref.shape[1], ref.shape[2]
    if use_bn:
        ref = ref * sc.double() + sh.double()
    res = torch.randn(B, OH, OW, cout, generator=g) if use_res else None
    if use_res:
        res = res.bfloat16() if yb else res
        ref = ref + res.double()
    if use_relu:
        ref = ref.clamp_min(0)
    conv = PackedConv(w.cuda(), stride=stride, pad=pad, dil=dil, scale=None if sc is None else sc.cuda(),
                      shift=None if sh is None else sh.cuda(), relu=use_relu)
    cands = [(t, s) for t in (1, 2, 3, 4, 21, 24) for s in (1, 2, 3)]      # 21 / 24: walked m-tile first
    if conv.patch_ok:
        cands += [(TILE_PATCH, s) for s in (1, 2, 3)]
    scale_ref = max(1.0, ref.abs().max().item())
    xd = (x.bfloat16() if xb else x).cuda()
    odt = torch.bfloat16 if yb else torch.float32
    for t, s in cands:
        nk = cin // 32 if t == TILE_PATCH else conv.k_pad // 32
        if s > nk:
            continue
        out = torch.full((B, OH, OW, cout + y_extra), -7.0, device="cuda", dtype=odt)
        try:
            conv(xd, out, x_coff=x_coff, y_coff=y_coff, residual=None if res is None else res.cuda(), tile=t, split_k=s)
            launches += 1
        except Exception as e:
            bad += 1
            print("EXC", (B, cin, H, W, cout, k, stride, pad, dil), (t, s), (xb, yb), str(e)[:160])
            continue
        got = out.float().cpu().double()
        err = (got[..., y_coff:y_coff + cout] - ref).abs().max().item()
        guard_ok = (got[..., :y_coff] == -7).all() and (got[..., y_coff + cout:] == -7).all()
        tol = (2.0 ** -8 if yb else 2e-4) * scale_ref
        if err > tol or not guard_ok or not torch.isfinite(got).all():
            bad += 1
            print("FAIL", (B, cin, H, W, cout, k, stride, pad, dil), (t, s), (xb, yb), "err", err, "tol", tol, "guard", bool(guard_ok),
                  dict(bn=use_bn, res=use_res, relu=use_relu, x_coff=x_coff, y_coff=y_coff))
torch.cuda.synchronize()
print(f"fuzz done (bf16 activations): {N} shapes, {launches} launches, failures: {bad}")
sys.exit(1 if bad else 0)
