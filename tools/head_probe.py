#!/usr/bin/env python3
"""Dev probe: the two fused fp32 CenterHead kernels at the cfg-2 launch (36 branches, 70 outputs, 64-channel 256x256 map):
conv_wino_head_kernel (F(2x2), csrc/conv_wino.hip) against head_wino4_kernel (F(4x4), csrc/head_wino4.hip), each timed as a
hipGraph of 10 launches, and their difference."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgv3d_amd import hip_ops                            # noqa: E402
from sgv3d_amd.hip_ops import PackedConv                 # noqa: E402
from tools.vp_probe3 import graph_us                     # noqa: E402


def main():
    dev = torch.device("cuda")
    counts = []
    for nc in (1, 2, 2, 1, 2, 2):
        counts += [2, 1, 3, 2, 2, nc]
    nb, total = len(counts), sum(counts)
    g = torch.Generator().manual_seed(36)
    H = W = int(os.environ.get("HEAD_PROBE_SIZE", "256"))
    x = torch.randn(1, H, W, 64, generator=g).to(dev)
    w1 = (torch.randn(nb * 64, 64, 3, 3, generator=g) / 24.0).to(dev)
    sc = (torch.rand(nb * 64, generator=g) + 0.5).to(dev)
    sh = (torch.randn(nb * 64, generator=g) * 0.2).to(dev)
    w2 = (torch.randn(total, 64, 3, 3, generator=g) / 24.0).permute(0, 2, 3, 1).contiguous().to(dev)
    b2 = torch.randn(total, generator=g).to(dev)
    ob = torch.tensor([0] + list(np.cumsum(counts)), dtype=torch.int32, device=dev)
    first = PackedConv(w1, pad=1, scale=sc, shift=sh, relu=True)
    u = hip_ops.pack_centerhead_f4(w1)
    out2 = torch.empty(1, total, H, W, device=dev)
    out4 = torch.empty(1, total, H, W, device=dev)
    f2 = lambda: hip_ops.centerhead_branches(x, first, w2, b2, ob, nb, out=out2)
    f4 = lambda: hip_ops.centerhead_branches_f4(x, u, sc, sh, w2, b2, ob, nb, out=out4)
    f2(); f4()
    torch.cuda.synchronize()
    print(f"max |F(4x4) - F(2x2)| = {float((out4 - out2).abs().max()):.3e} of scale {float(out2.abs().max()):.2f}", flush=True)
    flops = 2.0 * H * W * (nb * 64 * 64 * 9 + total * 9 * 64)
    for name, fn in (("F(2x2) conv_wino_head", f2), ("F(4x4) head_wino4", f4)):
        t = graph_us(fn, reps=10)
        print(f"{name}: {t:.1f} us per launch pair (kernel + ring fix-up), {flops / t / 1e6:.1f} direct-form TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
