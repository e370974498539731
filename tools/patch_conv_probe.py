"""Time the bf16 LDS-resident-patch 3x3 kernel against the implicit-GEMM bf16 tiles on the layer shapes of the bf16 configs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgv3d_amd import hip_ops

hip_ops.MFMA_BF16 = True
DEV = "cuda:0"
SHAPES = [  # B, cin, H, W, cout   (cfg-3: HeightNet 512 @ 68x120 x4, trunk 160/320/640, ResNet-101 3x3s)
    (4, 512, 68, 120, 512), (4, 256, 68, 120, 256), (4, 64, 272, 480, 64), (4, 128, 136, 240, 128), (4, 256, 68, 120, 256),
    (4, 512, 34, 60, 512), (4, 160, 256, 256, 160), (4, 320, 128, 128, 320), (4, 640, 64, 64, 640), (4, 64, 256, 256, 64),
    (1, 512, 108, 192, 512), (1, 256, 108, 192, 256),
]


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B, cin, H, W, cout in SHAPES:
    w = torch.randn(cout, cin, 3, 3, device=DEV) / (cin * 9) ** 0.5
    conv = hip_ops.PackedConv(w, stride=1, pad=1, scale=torch.ones(cout, device=DEV), shift=torch.zeros(cout, device=DEV), relu=True)
    flops = 2.0 * B * H * W * cout * cin * 9
    for xdt in (torch.bfloat16, torch.float32):
        x = torch.randn(B, H, W, cin, device=DEV).to(xdt)
        odt = xdt
        out = torch.empty(B, H, W, cout, dtype=odt, device=DEV)
        res = {}
        for t in (1, 2, 3, 4, hip_ops.TILE_PATCH):
            us = timeit(lambda: conv(x, out, tile=t, split_k=1))
            res[t] = us
        best_ig = min(res[t] for t in (1, 2, 3, 4))
        print(f"{B}x{H}x{W} {cin}->{cout} {str(xdt)[6:]:>8}: igemm best {best_ig:7.1f} us ({flops / best_ig / 1e6:6.0f} TF)   "
              f"patch {res[7]:7.1f} us ({flops / res[7] / 1e6:6.0f} TF)", flush=True)
