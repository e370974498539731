#!/usr/bin/env python3
"""Driver for counter runs of the all-taps bf16 weight gradient (tools/pmc_wgrad3x3_bf16.sh): a few launches of
conv_wgrad3x3_bf16_kernel on 512 -> 512 @54x96 (batch 2) and of the 36 CenterHead layers batched, nothing else on the device."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgv3d_amd import conv_grad, hip_ops

g = torch.Generator().manual_seed(0)
x = torch.randn(2, 54, 96, 512, generator=g).cuda()
dy = torch.randn(2, 54, 96, 512, generator=g).cuda()
for _ in range(6):
    conv_grad.conv2d_backward_weight_bf16(x, dy, 3, 1, 1, 1, tile=6, split=1)
hip_ops.MFMA_BF16, hip_ops.BF16_ACTIVATIONS, hip_ops.WGRAD_BF16_ALLTAPS = True, False, True
xs = torch.randn(2, 256, 256, 64, generator=g).cuda()
dys = [torch.randn(2, 256, 256, 64, generator=g).cuda() for _ in range(36)]
for _ in range(3):
    conv_grad.conv2d_backward_weight_batched(xs, dys)
torch.cuda.synchronize()
