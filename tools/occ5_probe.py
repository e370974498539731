#!/usr/bin/env python3
"""Dev probe: the five-workgroups-per-CU form of the f32 64x64 implicit-GEMM tile (host tiles 44 / 45, SGV3D_TILE_OCC5) against the
plain tile (4 / 24) on cfg-2 layers, each as a hipGraph of 10 launches -- alone, and as three concurrent copies."""
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
    layers = (("64->64 1x1 @216x384", 64, 64, 216, 384, 1, 1, 0, False), ("64->256 1x1 @216x384 +res", 64, 256, 216, 384, 1, 1, 0, True),
              ("256->64 1x1 @216x384", 256, 64, 216, 384, 1, 1, 0, False), ("256->128 1x1 @216x384", 256, 128, 216, 384, 1, 1, 0, False),
              ("128->512 1x1 @108x192 +res", 128, 512, 108, 192, 1, 1, 0, True), ("512->128 1x1 @108x192", 512, 128, 108, 192, 1, 1, 0, False),
              ("256->1024 1x1 @54x96 +res", 256, 1024, 54, 96, 1, 1, 0, True), ("1024->256 1x1 @54x96", 1024, 256, 54, 96, 1, 1, 0, False),
              ("512->2048 1x1 @27x48 +res", 512, 2048, 27, 48, 1, 1, 0, True), ("128->128 3x3 s2 @216x384", 128, 128, 216, 384, 3, 2, 1, False),
              ("1152->128 1x1 @54x96", 1152, 128, 54, 96, 1, 1, 0, False), ("2560->512 1x1 @54x96", 2560, 512, 54, 96, 1, 1, 0, False))
    for name, cin, cout, H, W, k, stride, pad, with_res in layers:
        x = torch.randn(1, H, W, cin, generator=g).to(dev)
        w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dev)
        sc, sh = (torch.rand(cout, generator=g) + 0.5).to(dev), (torch.randn(cout, generator=g) * 0.2).to(dev)
        conv = PackedConv(w, stride=stride, pad=pad, scale=sc, shift=sh, relu=True)
        oh, ow = conv.out_hw(H, W)
        res = torch.randn(1, oh, ow, cout, generator=g).to(dev) if with_res else None
        out = torch.empty(1, oh, ow, cout, device=dev)
        flops = 2.0 * oh * ow * cin * cout * k * k
        line = []
        for t in (4, 24, 44, 45):
            if t in (24, 45) and cout <= 64:
                continue
            best = None
            for sk in (1, 2, 3):
                if sk > 1 and (conv.k_pad // 32 // sk < 4 or oh * ow // 64 * (cout // 64) * sk > 6144):
                    continue
                fn = lambda: conv(x, out=out, residual=res, tile=t, split_k=sk)
                fn()
                us = graph_us(fn, reps=10)
                if best is None or us < best[0]:
                    best = (us, sk)
            line.append(f"tile {t}: {best[0]:6.1f} us (split {best[1]}, {flops / best[0] / 1e6 / 157.3:.2f})")
        print(f"{name:28s} " + " | ".join(line), flush=True)


if __name__ == "__main__":
    main()
