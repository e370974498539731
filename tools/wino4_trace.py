import os, sys, torch
sys.path.insert(0, "/root/repo")
from sgv3d_amd.hip_ops import PackedConv
for cin, H, W, cout in [(512, 54, 96, 512), (640, 32, 32, 640), (256, 54, 96, 256)]:
    x = torch.randn(1, H, W, cin, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
    conv = PackedConv(w, pad=1, relu=True)
    out = torch.empty(1, H, W, cout, device="cuda")
    for _ in range(30):
        conv(x, out, tile=9, split_k=1)
    torch.cuda.synchronize()
