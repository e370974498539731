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
    import torch.nn.functional as F
    shapes = tuple(s + (1, 1, 0) for s in shapes) + (
        ("128->128 k3s2 @216x384", 128, 128, 216, 384, False, 3, 2, 1), ("256->256 k3s2 @108x192", 256, 256, 108, 192, False, 3, 2, 1),
        ("512->512 k3s2 @54x96", 512, 512, 54, 96, False, 3, 2, 1), ("256->512 k1s2 @216x384", 256, 512, 216, 384, False, 1, 2, 0),
        ("512->1024 k1s2 @108x192", 512, 1024, 108, 192, False, 1, 2, 0), ("1024->2048 k1s2 @54x96", 1024, 2048, 54, 96, False, 1, 2, 0),
        ("160->320 k3s2 @128x128", 160, 320, 128, 128, False, 3, 2, 1), ("320->640 k3s2 @64x64", 320, 640, 64, 64, False, 3, 2, 1),
        ("256->128 k4s4 @216x384", 256, 128, 216, 384, False, 4, 4, 0), ("512->128 k2s2 @108x192", 512, 128, 108, 192, False, 2, 2, 0),
        ("4->64 k7s2 @864x1536", 4, 64, 864, 1536, False, 7, 2, 3), ("80->160 k7s2 @256x256", 80, 160, 256, 256, False, 7, 2, 3))
    for name, cin, cout, H, W, with_res, k, st, pd in shapes:
        if only and only not in name:
            continue
        x = torch.randn(1, H, W, cin, generator=g).to(dev)
        w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dev)
        sc, sh = (torch.rand(cout, generator=g) + 0.5).to(dev), (torch.randn(cout, generator=g) * 0.2).to(dev)
        conv = PackedConv(w, stride=st, pad=pd, scale=sc, shift=sh, relu=True)
        OH, OW = conv.out_hw(H, W)
        res = torch.randn(1, OH, OW, cout, generator=g).to(dev) if with_res else None
        out = torch.empty(1, OH, OW, cout, device=dev)
        want = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), None, st, pd).permute(0, 2, 3, 1) * sc.double() + sh.double()
        if with_res:
            want = want + res.double()
        want = torch.relu(want)
        scale = float(want.abs().max())
        row = []
        for t, sk in ((44, 1), (45, 1), (44, 2), (44, 3), (44, 6), (4, 1), (4, 2), (24, 1), (1, 1), (2, 1)) + tuple((t, sk) for t in hip_ops.PW_X3_TILES for sk in (1, 2, 3, 4, 6)):
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
              " ".join(f"{t}/sk{k}:{us:.1f}" for us, t, k, _ in x3[:5]) + f" | worst x3 err {max(r[3] for r in x3):.1e}", flush=True)


if __name__ == "__main__":
    main()
