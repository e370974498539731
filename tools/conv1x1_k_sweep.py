"""Time of the bf16-in / bf16-out 1x1 layer against K at fixed M x N: the intercept is what a workgroup spends outside its k loop."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgv3d_amd import hip_ops

hip_ops.MFMA_BF16 = True
DEV = "cuda:0"
B, H, W = 4, 68, 120


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for cout in (1024, 256):
    for res in (True, False):
        row = []
        for cin in (64, 128, 256, 512, 1024, 2048):
            w = torch.randn(cout, cin, 1, 1, device=DEV) / cin ** 0.5
            conv = hip_ops.PackedConv(w, scale=torch.ones(cout, device=DEV), shift=torch.zeros(cout, device=DEV), relu=True)
            x = torch.randn(B, H, W, cin, device=DEV).bfloat16()
            r = torch.randn(B, H, W, cout, device=DEV).bfloat16() if res else None
            out = torch.empty(B, H, W, cout, dtype=torch.bfloat16, device=DEV)
            best = min((timeit(lambda: conv(x, out, residual=r, tile=t, split_k=1)), t) for t in (1, 2, 3, 4))
            mb = (B * H * W * (cin + cout * (2 if res else 1)) * 2) / 1e6
            row.append(f"K={cin}: {best[0]:.1f} us (tile {best[1]}, {mb / best[0]:.2f} TB/s)")
        print(f"N={cout} residual={res}: " + "  ".join(row), flush=True)
