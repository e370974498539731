#!/usr/bin/env python3
"""Race / determinism soak for the Winograd kernels: many shapes x repeats, every result compared bitwise with the
first run of the same configuration and against the implicit GEMM within fp32 tolerance."""
import itertools, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgv3d_amd.hip_ops import PackedConv, TILE_WINO, TILE_WINO_RES

torch.manual_seed(0)
bad = 0
shapes = [(1, 64, 54, 96, 128), (2, 32, 33, 65, 96), (1, 512, 27, 48, 256), (1, 160, 64, 64, 160), (1, 64, 128, 128, 256),
          (3, 8, 17, 40, 64), (1, 256, 54, 96, 256)]
for (B, cin, H, W, cout) in shapes:
    x = torch.randn(B, H, W, cin, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
    res = torch.randn(B, H, W, cout, device="cuda")
    conv = PackedConv(w, pad=1, scale=torch.rand(cout, device="cuda") + 0.5, shift=torch.randn(cout, device="cuda"), relu=True)
    ref = conv(x, residual=res, tile=4, split_k=1)
    for tile, sk in itertools.product((TILE_WINO, TILE_WINO_RES), (1, 2, 3)):
        if tile == TILE_WINO_RES and (cin > 96 or sk > 1):
            continue
        if cin // 8 < sk:
            continue
        first = None
        for rep in range(25):
            y = conv(x, residual=res, tile=tile, split_k=sk)
            if first is None:
                first = y.clone()
                err = (y - ref).abs().max().item()
                if err > 2e-4 * max(1.0, ref.abs().max().item()):
                    bad += 1
                    print("MISMATCH vs igemm", (B, cin, H, W, cout), tile, sk, err)
            elif not torch.equal(y, first):
                bad += 1
                print("NON-DETERMINISTIC", (B, cin, H, W, cout), tile, sk, rep, (y - first).abs().max().item())
                break
torch.cuda.synchronize()
print("stress done, failures:", bad)
sys.exit(1 if bad else 0)
