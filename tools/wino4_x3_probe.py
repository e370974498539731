#!/usr/bin/env python3
"""Dev probe: the three-launch F(4x4) path with its position GEMM on the f32 MFMA (tiles 46 / 47) and as f32x3 on the bf16 matrix
cores (tiles 50-59, csrc/gemm_x3_grouped.hip) on the cfg-2 layers that use it: time (a hipGraph of 10 launches) and the error of
each against a float64 convolution on the CPU-free path (torch float64 on the device)."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgv3d_amd import hip_ops                            # noqa: E402
from sgv3d_amd.hip_ops import PackedConv                 # noqa: E402
from tools.vp_probe3 import graph_us                     # noqa: E402


def main():
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(0)
    only = os.environ.get("ONLY")
    for name, cin, cout, H, W, dil in (("128->128 @108x192", 128, 128, 108, 192, 1), ("256->256 @54x96", 256, 256, 54, 96, 1),
                                       ("512->512 @27x48", 512, 512, 27, 48, 1), ("512->512 @54x96", 512, 512, 54, 96, 1),
                                       ("512->512 @54x96 d6", 512, 512, 54, 96, 6), ("512->512 @54x96 d12", 512, 512, 54, 96, 12),
                                       ("512->512 @54x96 d18", 512, 512, 54, 96, 18), ("160->160 @128x128", 160, 160, 128, 128, 1),
                                       ("320->320 @64x64", 320, 320, 64, 64, 1), ("640->640 @32x32", 640, 640, 32, 32, 1)):
        if only and (only != name if os.environ.get("EXACT") else only not in name):
            continue
        x = torch.randn(1, H, W, cin, generator=g).to(dev)
        w = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).to(dev)
        sc, sh = (torch.rand(cout, generator=g) + 0.5).to(dev), (torch.randn(cout, generator=g) * 0.2).to(dev)
        conv = PackedConv(w, pad=dil, dil=dil, scale=sc, shift=sh, relu=True)
        out = torch.empty(1, H, W, cout, device=dev)
        want = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), None, 1, dil, dil)
        want = torch.relu(want * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]).permute(0, 2, 3, 1)
        scale = float(want.abs().max())
        res = []
        for t in tuple(int(v) for v in os.environ["TILES"].split(",")) if os.environ.get("TILES") else (46, 47) + tuple(range(50, 60)):
            try:
                fn = lambda: conv(x, out=out, tile=t, split_k=1)
                fn()
            except Exception as e:
                res.append(f"{t}: n/a ({str(e)[:40]})" if os.environ.get("VERBOSE") else f"{t}: n/a")
                continue
            torch.cuda.synchronize()
            err = float((out.double() - want).abs().max()) / scale
            res.append(f"{t}: {graph_us(fn, reps=10):6.1f} us e={err:.1e}")
        print(f"{name:22s} " + " | ".join(res), flush=True)


if __name__ == "__main__":
    main()
