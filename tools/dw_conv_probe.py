"""Time the bf16 direct-weight kernel (host tile ids 31-35) against the implicit-GEMM bf16io tiles and the patch kernel on the
layers of the bf16 configs (ResNet-101 at cfg-3 batch 4, HeightNet / ASPP, BEV trunk), alone and with three launches in
flight, with the HBM and MFMA floors of each layer beside them."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgv3d_amd import hip_ops

hip_ops.MFMA_BF16 = True
DEV = "cuda:0"
SHAPES = [  # B, cin, H, W, cout, k, stride, pad, dil, residual
    (4, 64, 272, 480, 256, 1, 1, 0, 1, True), (4, 256, 272, 480, 64, 1, 1, 0, 1, False),
    (4, 256, 272, 480, 512, 1, 2, 0, 1, False),
    (4, 128, 136, 240, 512, 1, 1, 0, 1, True), (4, 512, 136, 240, 128, 1, 1, 0, 1, False),
    (4, 256, 68, 120, 1024, 1, 1, 0, 1, True), (4, 1024, 68, 120, 256, 1, 1, 0, 1, False),
    (4, 1024, 68, 120, 2048, 1, 2, 0, 1, False), (4, 512, 34, 60, 2048, 1, 1, 0, 1, True), (4, 2048, 34, 60, 512, 1, 1, 0, 1, False),
    (4, 2560, 68, 120, 512, 1, 1, 0, 1, False),
    (4, 256, 68, 120, 256, 3, 1, 1, 1, False), (4, 512, 34, 60, 512, 3, 1, 1, 1, False), (4, 128, 272, 480, 128, 3, 2, 1, 1, False),
    (4, 256, 136, 240, 256, 3, 2, 1, 1, False), (4, 512, 68, 120, 512, 3, 2, 1, 1, False),
    (4, 512, 68, 120, 512, 3, 1, 1, 1, False), (4, 512, 68, 120, 512, 3, 1, 6, 6, False), (4, 512, 68, 120, 512, 3, 1, 18, 18, False),
    (4, 96, 512, 512, 160, 7, 2, 3, 1, False), (4, 160, 256, 256, 320, 3, 2, 1, 1, False), (4, 320, 128, 128, 640, 3, 2, 1, 1, False),
    (4, 160, 256, 256, 160, 3, 1, 1, 1, False), (4, 64, 272, 480, 64, 3, 1, 1, 1, False),
]
if os.environ.get("SHAPES"):
    SHAPES = [tuple(int(v) for v in s.split(",")) for s in os.environ["SHAPES"].split(";")]
    SHAPES = [s[:9] + (bool(s[9]),) for s in SHAPES]


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


STREAMS = [torch.cuda.Stream() for _ in range(3)]


def time3(fn, n=8):
    """three launches in flight: n back-to-back launches on each of three streams, time per launch"""
    fn(); torch.cuda.synchronize()
    cur = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(cur)
    for s in STREAMS:
        s.wait_event(e0)
        with torch.cuda.stream(s):
            for _ in range(n):
                fn()
        cur.wait_stream(s)
    e1.record(cur); e1.synchronize()
    return e0.elapsed_time(e1) / (3 * n) * 1e3


DW = (31, 32, 33, 34, 35)
print(f"{'layer':44} {'HBM us':>7} {'MFMA us':>7} | {'igemm':>7} {'(3x)':>7} | {'patch':>7} {'(3x)':>7} | "
      + " ".join(f"{'dw' + str(t):>7} {'(3x)':>7}" for t in DW))
for B, cin, H, W, cout, k, stride, pad, dil, with_res in SHAPES:
    w = torch.randn(cout, cin, k, k, device=DEV) / (cin * k * k) ** 0.5
    conv = hip_ops.PackedConv(w, stride=stride, pad=pad, dil=dil, scale=torch.ones(cout, device=DEV), shift=torch.zeros(cout, device=DEV), relu=True)
    oh, ow = conv.out_hw(H, W)
    M = B * oh * ow
    x = torch.randn(B, H, W, cin, device=DEV).bfloat16()
    out = torch.empty(B, oh, ow, cout, dtype=torch.bfloat16, device=DEV)
    res = torch.randn(B, oh, ow, cout, device=DEV).bfloat16() if with_res else None
    hbm = 2.0 * (B * H * W * cin + M * cout * (2 if with_res else 1)) / 8e12 * 1e6
    mfma = 2.0 * M * cout * cin * k * k / 2.5e15 * 1e6
    ig = {}
    for t in (1, 2, 3, 4, 21, 22, 23, 24):
        try:
            ig[t] = timeit(lambda: conv(x, out, residual=res, tile=t, split_k=1))
        except Exception:
            continue
    bt = min(ig, key=ig.get)
    b3 = time3(lambda: conv(x, out, residual=res, tile=bt, split_k=1))
    pcol = f"{'-':>7} {'-':>7}"
    if conv.patch_ok:
        pcol = (f"{timeit(lambda: conv(x, out, residual=res, tile=7, split_k=1)):7.1f} "
                f"{time3(lambda: conv(x, out, residual=res, tile=7, split_k=1)):7.1f}")
    cols = []
    for t in DW:
        us = timeit(lambda: conv(x, out, residual=res, tile=t, split_k=1))
        u3 = time3(lambda: conv(x, out, residual=res, tile=t, split_k=1))
        cols.append(f"{us:7.1f} {u3:7.1f}")
    name = f"{B}x{H}x{W} {cin}->{cout} k{k} s{stride} d{dil}{' +res' if with_res else ''}"
    print(f"{name:44} {hbm:7.1f} {mfma:7.1f} | {ig[bt]:7.1f} {b3:7.1f} | {pcol} | " + " ".join(cols) + f"   (igemm tile {bt})", flush=True)
