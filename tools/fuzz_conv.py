#!/usr/bin/env python3
"""Random-shape parity fuzz of the convolution entry points against torch (CPU, fp64 accumulate):
implicit GEMM (all tiles, split-K), Winograd (streaming, patch-resident), with BN fold / residual / ReLU /
gate / channel-slice input and output.  usage: fuzz_conv.py [N=300] [seed=0] [mode=f32|bf16|f32x3]
(bf16: the reference convolves the bf16-rounded operands, so only the summation order differs; Winograd is f32-only)"""
import os, random, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgv3d_amd import hip_ops
from sgv3d_amd.hip_ops import PackedConv, TILE_WINO, TILE_WINO_RES, TILE_WINO_HALF

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
MODE = sys.argv[3] if len(sys.argv) > 3 else "f32"
hip_ops.MFMA_BF16, hip_ops.MFMA_F32X3 = MODE == "bf16", MODE == "f32x3"
rng = random.Random(seed)
g = torch.Generator().manual_seed(seed)
bad = 0
for it in range(N):
    k = rng.choice([1, 3, 3, 3, 5, 7])
    stride = rng.choice([1, 1, 1, 2])
    dil = rng.choice([1, 1, 1, 2, 3]) if k == 3 else 1
    pad = rng.choice([0, (k // 2) * dil, (k // 2) * dil])
    cin = rng.choice([4, 8, 16, 32, 40, 64, 96, 128, 160, 256])
    cout = rng.choice([1, 3, 8, 20, 64, 70, 128, 200, 288])
    B = rng.choice([1, 1, 2, 3])
    H, W = rng.randint(7, 48), rng.randint(7, 70)
    if (H + 2 * pad - dil * (k - 1) - 1) < 0 or (W + 2 * pad - dil * (k - 1) - 1) < 0:
        continue
    x_extra, y_extra = rng.choice([0, 0, 8]), rng.choice([0, 0, 4])
    x = torch.randn(B, H, W, cin + x_extra, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    use_bn, use_res, use_relu, use_gate = (rng.random() < 0.6 for _ in range(4))
    sc = torch.rand(cout, generator=g) + 0.5 if use_bn else None
    sh = torch.randn(cout, generator=g) if use_bn else None
    x_coff = x_extra // 2 // 4 * 4
    xin = x[..., x_coff:x_coff + cin]
    rx, rw = (xin.bfloat16().double(), w.bfloat16().double()) if MODE == "bf16" else (xin.double(), w.double())
    ref = F.conv2d(rx.permute(0, 3, 1, 2), rw, stride=stride, padding=pad, dilation=dil).permute(0, 2, 3, 1)
    OH, OW = ref.shape[1], ref.shape[2]
    if use_bn:
        ref = ref * sc.double() + sh.double()
    res = torch.randn(B, OH, OW, cout, generator=g) if use_res else None
    if use_res:
        ref = ref + res.double()
    if use_relu:
        ref = ref.clamp_min(0)
    gate = torch.rand(B, cout, generator=g) if use_gate else None
    if use_gate:
        ref = ref * gate.double()[:, None, None, :]
    conv = PackedConv(w.cuda(), stride=stride, pad=pad, dil=dil, scale=None if sc is None else sc.cuda(),
                      shift=None if sh is None else sh.cuda(), relu=use_relu)
    cands = [(t, s) for t in (1, 2, 3, 4, 21, 22, 23, 24) for s in (1, 2, 3)]      # 21..24: the same tiles walked m-tile first
    if conv.k_order == 1 and MODE not in ("bf16", "f32x3"):
        cands += [(t, s) for t in (44, 45) for s in (1, 2, 3)]      # the 64x64 tile at five workgroups per CU (44; 45: m-tile first)
    if conv.w_wino is not None and MODE != "bf16":
        cands += [(TILE_WINO, 1), (TILE_WINO, 2), (TILE_WINO, 3), (TILE_WINO_HALF, 1), (TILE_WINO_HALF, 2), (TILE_WINO_HALF, 3)]
        if cin <= 96:
            cands.append((TILE_WINO_RES, 1))
    # round 6: the f32x3 kernels -- implicit GEMM (tiles 60-92 x split-K, csrc/conv_pw_x3.hip) and the F(4x4) position GEMM (50-59)
    if MODE == "f32" and not use_gate and cout % 4 == 0:
        if conv.pw_x3_ok():
            cands += [(t, s) for t in hip_ops.PW_X3_TILES for s in (1, 2, 3) if t < 80 or cout >= 256]
        if conv.wino4_ok():
            cands += [(t, 1) for t in hip_ops.WINO4_X3_TILES]
    scale_ref = max(1.0, ref.abs().max().item())
    for t, s in cands:
        nk = (cin * k * k + 31) // 32 if t in hip_ops.PW_X3_TILES else conv.k_pad // 32 if (t < TILE_WINO or t > 20) else cin // 8      # (Winograd variants: k-steps of 8 channels)
        if s > nk:
            continue
        out = torch.full((B, OH, OW, cout + y_extra), -7.0, device="cuda")
        y_coff = y_extra // 2 // 4 * 4 if cout % 4 == 0 else 0
        if cout % 4 and y_extra:
            out = torch.full((B, OH, OW, cout), -7.0, device="cuda")
        try:
            conv(x.cuda(), out, x_coff=x_coff, y_coff=y_coff, residual=None if res is None else res.cuda(),
                 gate=None if gate is None else gate.cuda(), tile=t, split_k=s)
        except Exception as e:
            bad += 1
            print("EXC", (B, cin, H, W, cout, k, stride, pad, dil), (t, s), str(e)[:120])
            continue
        got = out.cpu().double()
        err = (got[..., y_coff:y_coff + cout] - ref).abs().max().item()
        guard_ok = (got[..., :y_coff] == -7).all() and (got[..., y_coff + cout:] == -7).all()
        if err > 2e-4 * scale_ref or not guard_ok or not torch.isfinite(got).all():
            bad += 1
            print("FAIL", (B, cin, H, W, cout, k, stride, pad, dil), (t, s), "err", err, "guard", bool(guard_ok),
                  dict(bn=use_bn, res=use_res, relu=use_relu, gate=use_gate, x_coff=x_coff, y_coff=y_coff))
torch.cuda.synchronize()
print(f"fuzz done ({MODE}): {N} shapes, failures: {bad}")
sys.exit(1 if bad else 0)
