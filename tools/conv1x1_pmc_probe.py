"""Run the bf16-in / bf16-out 1x1 layers of ResNet-101 layer 3 alone (for rocprofv3 --pmc ... -- python3 tools/conv1x1_pmc_probe.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgv3d_amd import hip_ops

hip_ops.MFMA_BF16 = True
DEV = "cuda:0"
B, H, W = 4, 68, 120
for cin, cout, res in [(1024, 256, False), (256, 1024, True)]:
    w = torch.randn(cout, cin, 1, 1, device=DEV) / cin ** 0.5
    conv = hip_ops.PackedConv(w, scale=torch.ones(cout, device=DEV), shift=torch.zeros(cout, device=DEV), relu=True)
    x = torch.randn(B, H, W, cin, device=DEV).bfloat16()
    r = torch.randn(B, H, W, cout, device=DEV).bfloat16() if res else None
    out = torch.empty(B, H, W, cout, dtype=torch.bfloat16, device=DEV)
    for t in (1, 2, 3, 4):
        for _ in range(int(os.environ.get("REPS", "6"))):
            conv(x, out, residual=r, tile=t, split_k=1)
    torch.cuda.synchronize()
