"""A few launches of the bf16 direct-weight kernel per shape / tile for `rocprofv3 --pmc` (tools/pmc_dw.sh)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgv3d_amd import hip_ops

hip_ops.MFMA_BF16 = True
DEV = "cuda:0"
CASES = [  # B, cin, H, W, cout, k, stride, pad, dil, residual, tile
    (4, 512, 68, 120, 512, 3, 1, 1, 1, False, 34), (4, 512, 68, 120, 512, 3, 1, 1, 1, False, 31),
    (4, 256, 68, 120, 256, 3, 1, 1, 1, False, 34), (4, 256, 68, 120, 1024, 1, 1, 0, 1, True, 31),
    (4, 1024, 68, 120, 256, 1, 1, 0, 1, False, 31), (4, 64, 272, 480, 256, 1, 1, 0, 1, True, 31),
    (4, 2560, 68, 120, 512, 1, 1, 0, 1, False, 34),
]
for B, cin, H, W, cout, k, stride, pad, dil, with_res, tile in CASES:
    w = torch.randn(cout, cin, k, k, device=DEV) / (cin * k * k) ** 0.5
    conv = hip_ops.PackedConv(w, stride=stride, pad=pad, dil=dil, scale=torch.ones(cout, device=DEV), shift=torch.zeros(cout, device=DEV), relu=True)
    oh, ow = conv.out_hw(H, W)
    x = torch.randn(B, H, W, cin, device=DEV).bfloat16()
    out = torch.empty(B, oh, ow, cout, dtype=torch.bfloat16, device=DEV)
    res = torch.randn(B, oh, ow, cout, device=DEV).bfloat16() if with_res else None
    for _ in range(6):
        conv(x, out, residual=res, tile=tile, split_k=1)
    torch.cuda.synchronize()
