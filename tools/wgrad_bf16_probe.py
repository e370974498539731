#!/usr/bin/env python3
"""Dev probe: weight gradients of cfg-2 training layers (batch 2) on the f32 MFMA kernel (measured best tile / split) against
the bf16-MFMA kernel (sgv3d_conv2d_backward_weight_bf16, tiles 64x64 and 128x128, a few splits), hipGraph of 5 launches each."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgv3d_amd import conv_grad                          # noqa: E402
from tools.vp_probe3 import graph_us                     # noqa: E402


def main():
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(0)
    B = 2
    for name, cin, cout, H, W, k, s, p, d in (("512->512 3x3 @54x96", 512, 512, 54, 96, 3, 1, 1, 1), ("256->256 3x3 @54x96", 256, 256, 54, 96, 3, 1, 1, 1),
                                               ("64->64 3x3 @216x384", 64, 64, 216, 384, 3, 1, 1, 1), ("64->256 1x1 @216x384", 64, 256, 216, 384, 1, 1, 0, 1),
                                               ("256->1024 1x1 @54x96", 256, 1024, 54, 96, 1, 1, 0, 1), ("2560->512 1x1 @54x96", 2560, 512, 54, 96, 1, 1, 0, 1),
                                               ("512->512 3x3 d12 @54x96", 512, 512, 54, 96, 3, 1, 12, 12), ("160->160 3x3 @128x128", 160, 160, 128, 128, 3, 1, 1, 1),
                                               ("256->64 3x3 @256x256", 256, 64, 256, 256, 3, 1, 1, 1), ("80->160 7x7 s2 @256x256", 80, 160, 256, 256, 7, 2, 3, 1)):
        oh = (H + 2 * p - d * (k - 1) - 1) // s + 1
        ow = (W + 2 * p - d * (k - 1) - 1) // s + 1
        x = torch.randn(B, H, W, cin, generator=g).to(dev)
        dy = torch.randn(B, oh, ow, cout, generator=g).to(dev)
        flop = 2.0 * B * oh * ow * cout * cin * k * k
        f32 = lambda: conv_grad.conv2d_backward_weight(x, dy, k, s, p, d)
        ref = f32()
        t32 = graph_us(f32, reps=5)
        res = []
        best = None
        for tile in (1, 4):
            for split in (0, 2, 4, 8, 16, 32):
                try:
                    fn = lambda: conv_grad.conv2d_backward_weight_bf16(x, dy, k, s, p, d, tile=tile, split=split)
                    got = fn()
                except Exception as e:
                    continue
                t = graph_us(fn, reps=5)
                if best is None or t < best[0]:
                    best = (t, tile, split, float((got - ref).abs().max() / ref.abs().max()))
        line = (f"{name:26s} f32 {t32:8.1f} us = {flop / t32 / 1e6:6.1f} TF | bf16 per tap {best[0]:8.1f} us = {flop / best[0] / 1e6:7.1f} TF "
                f"(tile {best[1]}, split {best[2]}; rel. diff to f32 {best[3]:.1e})")
        if k == 3 and s == 1:                                   # all nine taps per workgroup (tile 6; split = row chunks per column)
            at = None
            for split in (0, 1, 2, 4, 8):
                fn = lambda: conv_grad.conv2d_backward_weight_bf16(x, dy, k, s, p, d, tile=6, split=split)
                got = fn()
                t = graph_us(fn, reps=5)
                if at is None or t < at[0]:
                    at = (t, split, float((got - ref).abs().max() / ref.abs().max()))
            line += f" | all taps {at[0]:8.1f} us = {flop / at[0] / 1e6:7.1f} TF (chunks {at[1]}; {at[2]:.1e})"
        print(line, flush=True)
    # the 36 first layers of the CenterHead branches as one batched launch
    from sgv3d_amd import hip_ops
    hip_ops.MFMA_BF16, hip_ops.BF16_ACTIVATIONS = True, False
    x = torch.randn(B, 256, 256, 64, generator=g).to(dev)
    dys = [torch.randn(B, 256, 256, 64, generator=g).to(dev) for _ in range(36)]
    flop = 36 * 2.0 * B * 256 * 256 * 64 * 64 * 9
    for flag in (False, True):
        hip_ops.WGRAD_BF16_ALLTAPS = flag
        t = graph_us(lambda: conv_grad.conv2d_backward_weight_batched(x, dys), reps=3)
        print(f"36 x 64->64 3x3 @256x256 batched, {'all taps' if flag else 'per tap'}: {t:8.1f} us = {flop / t / 1e6:7.1f} TF", flush=True)


if __name__ == "__main__":
    main()
