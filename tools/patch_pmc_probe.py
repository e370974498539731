"""Run the bf16 patch kernel alone on a few layer shapes (for `rocprofv3 --pmc ... --kernel-trace -- python3 tools/patch_pmc_probe.py`)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgv3d_amd import hip_ops

hip_ops.MFMA_BF16 = True
DEV = "cuda:0"
SHAPES = [(4, 320, 128, 128, 320), (4, 160, 256, 256, 160), (4, 512, 68, 120, 512), (4, 64, 272, 480, 64)]
for B, cin, H, W, cout in SHAPES:
    w = torch.randn(cout, cin, 3, 3, device=DEV) / (cin * 9) ** 0.5
    conv = hip_ops.PackedConv(w, stride=1, pad=1, scale=torch.ones(cout, device=DEV), shift=torch.zeros(cout, device=DEV), relu=True)
    x = torch.randn(B, H, W, cin, device=DEV).bfloat16()
    out = torch.empty(B, H, W, cout, dtype=torch.bfloat16, device=DEV)
    for _ in range(int(os.environ.get("REPS", "10"))):
        conv(x, out, tile=hip_ops.TILE_PATCH, split_k=1)
    torch.cuda.synchronize()
