"""Fused conv2 + conv3 of a 256-channel bottleneck (sgv3d_conv_dw_bf16_pair_forward) against the two launches with their best
tiles: alone and with three launches in flight."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sgv3d_amd import hip_ops

hip_ops.MFMA_BF16 = True
DEV = "cuda:0"
STREAMS = [torch.cuda.Stream() for _ in range(3)]


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def time3(fn, n=8):
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


for B, H, W, stride in ((4, 68, 120, 1), (4, 136, 240, 2), (1, 54, 96, 1), (4, 54, 96, 1), (1, 108, 192, 2)):
    cin = 256
    ca = hip_ops.PackedConv(torch.randn(256, cin, 3, 3, device=DEV) / 48, stride=stride, pad=1, scale=torch.ones(256, device=DEV),
                            shift=torch.zeros(256, device=DEV), relu=True)
    cb = hip_ops.PackedConv(torch.randn(1024, 256, 1, 1, device=DEV) / 16, scale=torch.ones(1024, device=DEV), shift=torch.zeros(1024, device=DEV), relu=True)
    x = torch.randn(B, H, W, cin, device=DEV).bfloat16()
    oh, ow = ca.out_hw(H, W)
    res = torch.randn(B, oh, ow, 1024, device=DEV).bfloat16()
    mid = torch.empty(B, oh, ow, 256, dtype=torch.bfloat16, device=DEV)
    out = torch.empty(B, oh, ow, 1024, dtype=torch.bfloat16, device=DEV)
    best = None
    for ta in (31, 34, 32, 7, 21, 36, 38, 39):
        for tb in (31, 34, 32, 36, 38, 39):
            try:
                two = lambda: cb(ca(x, mid, tile=ta, split_k=1), out, residual=res, tile=tb, split_k=1)
                us = timeit(two)
            except Exception:
                continue
            if best is None or us < best[0]:
                best = (us, ta, tb)
    ta, tb = best[1], best[2]
    two = lambda: cb(ca(x, mid, tile=ta, split_k=1), out, residual=res, tile=tb, split_k=1)
    one = lambda: hip_ops.conv_pair_bf16(ca, cb, x, res, out)
    print(f"{B}x{H}x{W} 256->256 k3 s{stride} ->1024 +res: two launches {timeit(two):7.1f} us ({time3(two):7.1f} 3x; tiles {ta},{tb})   fused {timeit(one):7.1f} us ({time3(one):7.1f} 3x)", flush=True)
