#!/usr/bin/env python3
"""Dev probe: the 1x1 layers of cfg-2 on the f32 MFMA (five-per-CU tiles 44 / 45, the committed choices) and as pointwise f32x3
(tiles 60-76, csrc/conv_pw_x3.hip): time (a hipGraph of 10 launches) and error against a float64 product."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgv3d_amd import hip_ops                            # noqa: E402
from sgv3d_amd.hip_ops import PackedConv                 # noqa: E402
from tools.vp_probe3 import graph_us                     # noqa: E402


def main():
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(0)
    only = os.environ.get("ONLY")
    shapes = (("64->256 @216x384 +res", 64, 256, 216, 384, True), ("256->64 @216x384", 256, 64, 216, 384, False),
              ("128->512 @108x192 +res", 128, 512, 108, 192, True), ("512->128 @108x192", 512, 128, 108, 192, False),
              ("256->1024 @54x96 +res", 256, 1024, 54, 96, True), ("1024->256 @54x96", 1024, 256, 54, 96, False),
              ("512->2048 @27x48 +res", 512, 2048, 27, 48, True), ("2048->512 @27x48", 2048, 512, 27, 48, False),
              ("2048->512 @54x96", 2048, 512, 54, 96, False), ("512->512 @54x96", 512, 512, 54, 96, False),
              ("1024->512 @54x96", 1024, 512, 54, 96, False), ("512->256 @108x192", 512, 256, 108, 192, False))
    for name, cin, cout, H, W, with_res in shapes:
        if only and only not in name:
            continue
        x = torch.randn(1, H, W, cin, generator=g).to(dev)
        w = (torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5).to(dev)
        sc, sh = (torch.rand(cout, generator=g) + 0.5).to(dev), (torch.randn(cout, generator=g) * 0.2).to(dev)
        res = torch.randn(1, H, W, cout, generator=g).to(dev) if with_res else None
        conv = PackedConv(w, scale=sc, shift=sh, relu=True)
        out = torch.empty(1, H, W, cout, device=dev)
        want = x.double().reshape(-1, cin) @ w.double().reshape(cout, cin).t() * sc.double() + sh.double()
        if with_res:
            want = want + res.double().reshape(-1, cout)
        want = torch.relu(want).reshape(1, H, W, cout)
        scale = float(want.abs().max())
        row = []
        for t, sk in ((44, 1), (45, 1), (44, 3)) + tuple((t, 1) for t in hip_ops.PW_X3_TILES):
            try:
                fn = lambda: conv(x, out=out, residual=res, tile=t, split_k=sk)
                fn()
            except Exception as e:
                continue
            torch.cuda.synchronize()
            err = float((out.double() - want).abs().max()) / scale
            row.append((graph_us(fn, reps=10), t, sk, err))
        f32 = min(r for r in row if r[1] < 60)
        x3 = sorted(r for r in row if r[1] >= 60)
        print(f"{name:24s} f32 best {f32[0]:6.1f} us (tile {f32[1]} sk{f32[2]}, e={f32[3]:.1e}) | x3: " +
              " ".join(f"{t}:{us:.1f}" for us, t, _, _ in x3[:5]) + f" | worst x3 err {max(r[3] for r in x3):.1e}", flush=True)


if __name__ == "__main__":
    main()
