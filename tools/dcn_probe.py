#!/usr/bin/env python3
"""Dev probe: the deformable 3x3 convolution of HeightNet (lss_fpn.py:190-198; cfg-2: 512 -> 512, groups 4, 54x96) as
deform_im2col3x3 + one GEMM per group (rounds 1-4) against sgv3d_deform_conv3x3_forward (one launch), each as a hipGraph of 10."""
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
    for name, B, C, H, W, groups in (("cfg-2 512 g4 @54x96", 1, 512, 54, 96, 4), ("cfg-3 b4 512 g4 @68x120", 4, 512, 68, 120, 4)):
        cpg = C // groups
        x = torch.randn(B, H, W, C, generator=g).to(dev)
        off = (torch.randn(B, H, W, 18, generator=g) * 1.5).to(dev)
        weight = torch.randn(C, cpg, 3, 3, generator=g) / (9 * cpg) ** 0.5
        convs = [PackedConv(weight[gi * cpg:(gi + 1) * cpg].permute(0, 2, 3, 1).reshape(cpg, 9 * cpg, 1, 1).contiguous().to(dev))
                 for gi in range(groups)]
        out_a = torch.empty(B, H, W, C, device=dev)
        out_b = torch.empty(B, H, W, C, device=dev)

        def old():
            col = hip_ops.deform_im2col3x3(x, off, groups)
            for gi, conv in enumerate(convs):
                conv(col, out_a, x_coff=gi * 9 * cpg, y_coff=gi * cpg)

        def new():
            hip_ops.deform_conv3x3(x, off, convs, out=out_b)
        old(); new()
        torch.cuda.synchronize()
        err = float((out_a - out_b).abs().max()) / float(out_a.abs().max())
        flop = 2.0 * B * H * W * C * 9 * cpg
        t_old, t_new = graph_us(old, reps=10), graph_us(new, reps=10)
        print(f"{name:26s} im2col + {groups} GEMMs {t_old:7.1f} us | fused {t_new:7.1f} us = {flop / t_new / 1e6:6.1f} TFLOP/s "
              f"({flop / t_new / 1e6 / 157.3:.2f} of the f32 MFMA peak) | rel. difference {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
