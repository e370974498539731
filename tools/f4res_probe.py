#!/usr/bin/env python3
"""Dev probe: the resident F(4x4) convolution (conv_f4res_kernel, csrc/head_wino4.hip) against the kernels the tune DB holds
for the same layers of cfg-2 -- ResNet layer 1's 64 -> 64 @216x384 (F(2x2) half-position kernel) and the CenterHead's shared
256 -> 64 @256x256 (F(2x2) kernel) -- each as a hipGraph of 10 launches."""
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
    for name, (cin, cout, H, W), others in (("layer1 64->64 @216x384", (64, 64, 216, 384), (hip_ops.TILE_WINO_HALF, hip_ops.TILE_WINO)),
                                            ("shared 256->64 @256x256", (256, 64, 256, 256), (hip_ops.TILE_WINO, hip_ops.TILE_WINO_HALF, hip_ops.TILE_WINO4))):
        x = torch.randn(1, H, W, cin, generator=g).to(dev)
        w = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).to(dev)
        sc, sh = (torch.rand(cout, generator=g) + 0.5).to(dev), (torch.randn(cout, generator=g) * 0.2).to(dev)
        conv = PackedConv(w, pad=1, scale=sc, shift=sh, relu=True)
        out = torch.empty(1, H, W, cout, device=dev)
        ref = conv(x, tile=others[0], split_k=1).clone()
        flops = 2.0 * H * W * cin * cout * 9
        for t in (hip_ops.TILE_F4RES,) + others:
            try:
                fn = lambda: conv(x, out=out, tile=t, split_k=1)
                fn()
            except Exception as e:
                print(f"{name}: tile {t} not available ({e})")
                continue
            torch.cuda.synchronize()
            err = float((out - ref).abs().max())
            us = graph_us(fn, reps=10)
            print(f"{name}: tile {t:2d} ({hip_ops.TILE_NAMES[t]}): {us:6.1f} us, {flops / us / 1e6:6.1f} direct-form TFLOP/s, "
                  f"max |diff to tile {others[0]}| {err:.2e}", flush=True)


if __name__ == "__main__":
    main()
