#!/usr/bin/env python3
"""Dev probe: the three-launch F(4x4) path with the GEMM tiles 64x64 (9), 64x128 (10), 32x128 (15) and the five-per-CU 64x64 (46)
and the 16x16x4-MFMA 48x64 tile (47) on the cfg-2 layers that use it, each as a hipGraph of 10 launches."""
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
    for name, cin, cout, H, W, dil in (("128->128 @108x192", 128, 128, 108, 192, 1), ("256->256 @54x96", 256, 256, 54, 96, 1),
                                       ("512->512 @27x48", 512, 512, 27, 48, 1), ("512->512 @54x96", 512, 512, 54, 96, 1),
                                       ("512->512 @54x96 d6", 512, 512, 54, 96, 6), ("512->512 @54x96 d12", 512, 512, 54, 96, 12),
                                       ("512->512 @54x96 d18", 512, 512, 54, 96, 18), ("160->160 @128x128", 160, 160, 128, 128, 1),
                                       ("320->320 @64x64", 320, 320, 64, 64, 1), ("640->640 @32x32", 640, 640, 32, 32, 1)):
        x = torch.randn(1, H, W, cin, generator=g).to(dev)
        w = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).to(dev)
        sc, sh = (torch.rand(cout, generator=g) + 0.5).to(dev), (torch.randn(cout, generator=g) * 0.2).to(dev)
        conv = PackedConv(w, pad=dil, dil=dil, scale=sc, shift=sh, relu=True)
        out = torch.empty(1, H, W, cout, device=dev)
        res = []
        ref = None
        for t in (9, 10, 15, 46, 47):
            try:
                fn = lambda: conv(x, out=out, tile=t, split_k=1)
                fn()
            except Exception as e:
                res.append(f"tile {t}: n/a")
                continue
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            same = bool(torch.equal(out, ref)) if t != 47 else bool((out - ref).abs().max() <= 1e-4 * ref.abs().max())   # (47: another k order)
            res.append(f"tile {t}: {graph_us(fn, reps=10):6.1f} us{'' if same else ' (differs!)'}")
        print(f"{name:22s} " + " | ".join(res), flush=True)


if __name__ == "__main__":
    main()
