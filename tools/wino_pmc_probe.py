"""For rocprofv3 --pmc: the two Winograd variants on the HeightNet / trunk shapes of cfg-2."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgv3d_amd import hip_ops
DEV = "cuda:0"
for B, cin, H, W, cout, sk in [(1, 512, 54, 96, 512, 3), (1, 512, 54, 96, 512, 1), (1, 160, 128, 128, 160, 1)]:
    w = torch.randn(cout, cin, 3, 3, device=DEV) / (cin * 9) ** 0.5
    conv = hip_ops.PackedConv(w, pad=1, relu=True)
    x = torch.randn(B, H, W, cin, device=DEV)
    out = torch.empty(B, H, W, cout, device=DEV)
    for t in (hip_ops.TILE_WINO, hip_ops.TILE_WINO_HALF):
        for _ in range(6):
            conv(x, out, tile=t, split_k=sk)
    torch.cuda.synchronize()
