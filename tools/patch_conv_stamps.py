"""Where does a workgroup of conv_patch_bf16_kernel spend its cycles?  (cycle-counter stamps of workgroup 0)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgv3d_amd import _lib, hip_ops

hip_ops.MFMA_BF16 = True
lib = _lib.load()
DEV = "cuda:0"
for B, cin, H, W, cout in [(4, 320, 128, 128, 320), (4, 512, 68, 120, 512), (4, 64, 272, 480, 64)]:
    w = torch.randn(cout, cin, 3, 3, device=DEV) / (cin * 9) ** 0.5
    conv = hip_ops.PackedConv(w, stride=1, pad=1, scale=torch.ones(cout, device=DEV), shift=torch.zeros(cout, device=DEV), relu=True)
    x = torch.randn(B, H, W, cin, device=DEV).bfloat16()
    out = torch.empty(B, H, W, cout, dtype=torch.bfloat16, device=DEV)
    for _ in range(3):
        conv(x, out, tile=hip_ops.TILE_PATCH, split_k=1)
    n = cin // 32
    dbg = torch.zeros(5 + 2 * n, dtype=torch.int64, device=DEV)
    lib.sgv3d_conv3x3_patch_bf16_debug_stamps(dbg.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); conv(x, out, tile=hip_ops.TILE_PATCH, split_k=1); e1.record()
    torch.cuda.synchronize()
    lib.sgv3d_conv3x3_patch_bf16_debug_stamps(None)
    t = dbg.cpu().tolist()
    mf = [t[2 + 2 * i] - (t[1] if i == 0 else t[1 + 2 * i]) for i in range(n)]
    hand = [t[3 + 2 * i] - t[2 + 2 * i] for i in range(n)]
    print(f"{B}x{H}x{W} {cin}->{cout}: kernel {e0.elapsed_time(e1) * 1e3:.0f} us; workgroup 0 total {t[2 + 2 * n] - t[0]} ticks: "
          f"prologue {t[1] - t[0]}, stages (MFMA part) {mf}, hand-over (store + barrier) {hand}, epilogue {t[2 + 2 * n] - t[1 + 2 * n]} (channel terms + residual fetch {t[3 + 2 * n] - t[1 + 2 * n]}, first n-tile {t[4 + 2 * n] - t[3 + 2 * n]}, second {t[2 + 2 * n] - t[4 + 2 * n]}); "
          f"ideal per stage = 144 MFMAs x 32 = 4608 (x2 with two waves per SIMD)")
